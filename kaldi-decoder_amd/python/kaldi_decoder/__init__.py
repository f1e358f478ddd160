"""MI355X-native drop-in for the kaldi_decoder package.

Exports the reference's names
(/root/reference/kaldi-decoder/python/kaldi_decoder/__init__.py:1-9) from the
pybind11 module built in-tree, plus what replaces the kaldifst types the
reference borrows (Fst, Lattice, get_linear_symbol_sequence) and the batched
decoder.  There is no CPU fallback: importing without the built extension, or
decoding without a GPU, raises.

Submodules: `cabi` (ctypes binding of the C ABI), `synth` (seeded synthetic
graphs and log-probs), `sharded` (multi-GPU split / gather helpers).
"""
import os as _os

# PyTorch-ROCm bundles its own libamdhip64.so.7; load it first so the extension's
# DT_NEEDED entry binds to the same HIP runtime (two runtimes cannot share a GPU).
try:
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover
    _torch = None

try:
    from .lib._kaldi_decoder import (  # noqa: F401
        BatchFasterDecoder,
        BatchLatticeDecoder,
        DecodableCtc,
        DecodableInterface,
        DeviceOptions,
        FasterDecoder,
        FasterDecoderOptions,
        Fst,
        Lattice,
        LatticeFasterDecoder,
        LatticeFasterDecoderConfig,
        LatticeSimpleDecoder,
        LatticeSimpleDecoderConfig,
        SimpleDecoder,
        get_linear_symbol_sequence,
        shortest_path,
    )
except ImportError as e:  # pragma: no cover
    raise ImportError(
        "kaldi_decoder: the native extension is not built "
        f"({_os.path.join(_os.path.dirname(__file__), 'lib')}). Run "
        "`make -C kaldi-decoder_amd` or `python __graft_entry__.py`; there is no "
        f"CPU fallback. Original error: {e}") from e

__version__ = "0.1.0"


def fst_from_dict(g):
    """Fst from a CSR dictionary (see kaldi_decoder.synth): keys start, row_ptr,
    ilabel, olabel, weight, nextstate, final."""
    return Fst(int(g["start"]), g["row_ptr"], g["ilabel"], g["olabel"], g["weight"],
               g["nextstate"], g["final"])


def fst_from_arcs(num_states, start, arcs, finals):
    """Fst from (src, dst, ilabel, olabel, weight) tuples and {state: final cost}."""
    from . import synth
    return fst_from_dict(synth.graph_from_arcs(num_states, start, arcs, finals))


def decodable_from_tensor(log_probs, offset=0):
    """DecodableCtc over a 2-D float32 torch tensor without copying when it is a
    contiguous-row ROCm tensor (the search then reads it in place in HBM)."""
    t = log_probs
    if _torch is not None and isinstance(t, _torch.Tensor):
        if t.dim() != 2 or t.dtype != _torch.float32:
            raise ValueError("log_probs must be a 2-D float32 tensor")
        if t.is_cuda:
            if t.stride(1) != 1:
                t = t.contiguous()
            d = DecodableCtc.from_pointer(t.data_ptr(), t.shape[0], t.shape[1], offset,
                                          True, t.stride(0))
            d._keep_alive = t
            return d
        t = t.numpy()
    return DecodableCtc(t, offset)
