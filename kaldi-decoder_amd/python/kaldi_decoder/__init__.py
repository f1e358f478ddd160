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

__version__ = "0.2.0"


def as_fst(graph):
    """The decoding graph as this package's Fst.  The reference's constructors take
    kaldifst's fst::Fst / VectorFst / ConstFst<StdArc> objects
    (/root/reference/kaldi-decoder/python/csrc/faster-decoder.cc:34-42); anything that can
    write itself as an OpenFst binary file -- kaldifst.StdVectorFst, StdConstFst, a
    pywrapfst Fst: they all have .write(path) -- is taken through such a file
    (csrc/fst-io.cc reads the `vector` and `const` containers); a path is read directly."""
    if isinstance(graph, Fst):
        return graph
    if isinstance(graph, (str, _os.PathLike)):
        return Fst.read(_os.fspath(graph))
    if hasattr(graph, "write"):
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            path = _os.path.join(d, "graph.fst")
            graph.write(path)
            return Fst.read(path)
    raise TypeError(f"cannot use {type(graph).__name__} as a decoding graph: expected kaldi_decoder.Fst, "
                    "a path to an OpenFst binary file, or an object with .write(path)")


def _taking_any_graph(cls):
    """The decoder class with a constructor that accepts what as_fst accepts (and keeps
    the converted graph alive: the reference borrows its graph, faster-decoder.h:179)."""
    class _Decoder(cls):
        def __init__(self, fst, *args, **kwargs):
            fst = as_fst(fst)
            cls.__init__(self, fst, *args, **kwargs)
            self._graph = fst
    _Decoder.__name__ = cls.__name__
    _Decoder.__qualname__ = cls.__qualname__
    _Decoder.__doc__ = cls.__doc__
    return _Decoder


FasterDecoder = _taking_any_graph(FasterDecoder)
SimpleDecoder = _taking_any_graph(SimpleDecoder)
LatticeSimpleDecoder = _taking_any_graph(LatticeSimpleDecoder)
LatticeFasterDecoder = _taking_any_graph(LatticeFasterDecoder)
BatchFasterDecoder = _taking_any_graph(BatchFasterDecoder)
BatchLatticeDecoder = _taking_any_graph(BatchLatticeDecoder)


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
