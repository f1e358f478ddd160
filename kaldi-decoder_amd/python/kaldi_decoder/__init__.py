"""MI355X-native drop-in for the kaldi_decoder package.

Same names as /root/reference/kaldi-decoder/python/kaldi_decoder/__init__.py:1-9
once the pybind11 module is built; `cabi` is the plain C-ABI binding, `synth`
the seeded synthetic graphs, `sharded` the multi-GPU split/gather helpers.
"""
