"""ctypes binding of libkd_hip.so, the C ABI declared in include/kd_hip.h.

This is the "ctypes stub" form of the drop-in boundary: plain
pointers and sizes in, plain arrays out.  It never falls back to a CPU
implementation: if the library is missing or there is no GPU the calls raise.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", os.environ.get("KD_HIP_LIB", "libkd_hip.so"))

INT_MAX = 2**31 - 1
KD_MEM_HOST, KD_MEM_DEVICE = 0, 1
KD_DECODE_FINALIZE, KD_DECODE_LATTICE, KD_DECODE_BEST_PATH = 1, 2, 4
CAPACITY_NAMES = ("arena_tokens", "arena_links", "table_slots", "max_frames", "tokens_used",
                  "links_used", "grow_events", "workgroups")
STAT_NAMES = ("frames", "tokens_in", "tokens_expanded", "arcs_scanned", "arcs_kept",
              "tokens_out", "eps_arcs_scanned", "eps_expansions", "max_tokens", "spill_frames",
              "links", "small_frames", "arcs_skipped", "index_probes", "handovers", "small_wg_frames", "prunes")

# every symbol include/kd_hip.h declares
SYMBOLS = (
    "kd_last_error", "kd_device_count", "kd_graph_create", "kd_graph_destroy",
    "kd_graph_num_states", "kd_graph_num_arcs", "kd_graph_device_bytes",
    "kd_faster_options_default", "kd_batch_create", "kd_batch_destroy",
    "kd_batch_set_options", "kd_batch_init_decoding", "kd_batch_advance",
    "kd_batch_advance_async", "kd_batch_sync", "kd_batch_num_frames_decoded",
    "kd_batch_decode_host_async", "kd_batch_status", "kd_batch_reached_final", "kd_batch_best_path_prepare",
    "kd_batch_best_path_copy", "kd_batch_frame_tokens", "kd_batch_stats",
    "kd_batch_last_kernel_ms", "kd_batch_device_bytes", "kd_batch_phase_cycles",
    "kd_lattice_options_default", "kd_batch_create_lattice", "kd_batch_finalize_decoding",
    "kd_batch_final_relative_cost", "kd_batch_raw_lattice_prepare", "kd_batch_raw_lattice_copy",
    "kd_batch_create_simple", "kd_batch_decode_async", "kd_batch_decode", "kd_batch_capacity",
    "kd_batch_set_output_buffer", "kd_batch_raw_lattice_table", "kd_batch_set_log_softmax",
    "kd_batch_split_info", "kd_batch_set_split", "kd_batch_sched_profile", "kd_batch_set_lazy_prune",
)
SPLIT_NAMES = ("split", "large_cus", "large_workgroups", "small_workgroups", "mig_lo", "mig_hi",
               "prune_small", "min_batch")


class FasterOptions(C.Structure):
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("beam_delta", C.c_float), ("hash_ratio", C.c_float),
                ("arena_tokens", C.c_int64), ("table_slots", C.c_int32),
                ("max_frames", C.c_int32), ("no_grow", C.c_int32)]


class LatticeOptions(C.Structure):
    _fields_ = [("beam", C.c_float), ("max_active", C.c_int32), ("min_active", C.c_int32),
                ("lattice_beam", C.c_float), ("prune_interval", C.c_int32),
                ("beam_delta", C.c_float), ("hash_ratio", C.c_float), ("prune_scale", C.c_float),
                ("arena_tokens", C.c_int64), ("arena_links", C.c_int64),
                ("table_slots", C.c_int32), ("max_frames", C.c_int32), ("no_grow", C.c_int32)]


_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C kaldi-decoder_amd` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    # PyTorch-ROCm wheels bundle their own libamdhip64.so.7.  Two HIP runtimes in
    # one process cannot both own the GPU, so when torch is installed load it
    # first: libkd_hip.so's DT_NEEDED libamdhip64.so.7 then binds to that copy.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    L.kd_last_error.restype = C.c_char_p
    L.kd_graph_create.argtypes = [C.c_int, C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _f32p,
                                  _i32p, _f32p, C.POINTER(C.c_void_p)]
    L.kd_graph_destroy.argtypes = [C.c_void_p]
    L.kd_graph_destroy.restype = None
    L.kd_graph_num_states.argtypes = [C.c_void_p]
    L.kd_graph_num_arcs.argtypes = [C.c_void_p]
    L.kd_graph_num_arcs.restype = C.c_int64
    L.kd_graph_device_bytes.argtypes = [C.c_void_p]
    L.kd_graph_device_bytes.restype = C.c_size_t
    L.kd_faster_options_default.argtypes = [C.POINTER(FasterOptions)]
    L.kd_faster_options_default.restype = None
    L.kd_batch_create.argtypes = [C.c_void_p, C.POINTER(FasterOptions), C.c_int32, C.c_void_p,
                                  C.POINTER(C.c_void_p)]
    L.kd_batch_destroy.argtypes = [C.c_void_p]
    L.kd_batch_destroy.restype = None
    L.kd_batch_set_options.argtypes = [C.c_void_p, C.POINTER(FasterOptions)]
    L.kd_batch_init_decoding.argtypes = [C.c_void_p, C.c_int32]
    L.kd_batch_advance.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), _i32p, C.c_void_p,
                                   C.c_int32, C.c_int64, C.c_int, C.c_int32]
    L.kd_batch_advance_async.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), _i32p, C.c_void_p,
                                         C.c_int32, C.c_int64, C.c_int32]
    L.kd_batch_sync.argtypes = [C.c_void_p]
    L.kd_batch_decode_async.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), _i32p,
                                        C.c_void_p, C.c_int32, C.c_int64, C.c_int32]
    L.kd_batch_decode.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), _i32p, C.c_void_p,
                                  C.c_int32, C.c_int64, C.c_int, C.c_int32]
    L.kd_batch_decode_host_async.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), _i32p,
                                             C.c_void_p, C.c_int32, C.c_int64, C.c_int32]
    L.kd_batch_capacity.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    L.kd_batch_split_info.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int32)]
    L.kd_batch_set_lazy_prune.argtypes = [C.c_void_p, C.c_int32]
    L.kd_batch_sched_profile.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    L.kd_batch_set_split.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.kd_batch_num_frames_decoded.argtypes = [C.c_void_p, _i32p]
    L.kd_batch_status.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.uint32)]
    L.kd_batch_reached_final.argtypes = [C.c_void_p, _i32p]
    L.kd_batch_best_path_prepare.argtypes = [C.c_void_p, C.c_int32, _i32p]
    L.kd_batch_best_path_copy.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int64), _i32p,
                                          _i32p, _i32p, _f32p, _f32p, _f32p]
    L.kd_batch_frame_tokens.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                        C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    L.kd_batch_stats.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    L.kd_batch_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.kd_batch_device_bytes.argtypes = [C.c_void_p]
    L.kd_batch_device_bytes.restype = C.c_size_t
    L.kd_batch_phase_cycles.argtypes = [C.c_void_p, np.ctypeslib.ndpointer(np.int64)]
    L.kd_lattice_options_default.argtypes = [C.POINTER(LatticeOptions)]
    L.kd_lattice_options_default.restype = None
    L.kd_batch_create_lattice.argtypes = [C.c_void_p, C.POINTER(LatticeOptions), C.c_int32,
                                          C.c_void_p, C.POINTER(C.c_void_p)]
    L.kd_batch_create_simple.argtypes = [C.c_void_p, C.c_float, C.c_int64, C.c_int32, C.c_int32,
                                         C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    L.kd_batch_finalize_decoding.argtypes = [C.c_void_p]
    L.kd_batch_final_relative_cost.argtypes = [C.c_void_p, _f32p]
    L.kd_batch_raw_lattice_prepare.argtypes = [C.c_void_p, C.c_int32, _i32p]
    _i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    L.kd_batch_raw_lattice_copy.argtypes = [C.c_void_p, _i64p, _i64p, _i64p, _i32p, _i32p, _i32p,
                                            _i32p, _f32p, _f32p, _i32p, _i32p, _i32p, _f32p]
    L.kd_batch_set_output_buffer.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    L.kd_batch_set_log_softmax.argtypes = [C.c_void_p, C.c_int32]
    L.kd_batch_raw_lattice_table.argtypes = [C.c_void_p, C.POINTER(C.c_int64),
                                             np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise RuntimeError(lib().kd_last_error().decode())


def device_count():
    return lib().kd_device_count()


def default_options(**kw):
    o = FasterOptions()
    lib().kd_faster_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Graph:
    """Device copy of a decoding graph (kd_graph)."""

    def __init__(self, g, device=0):
        self.h = C.c_void_p()
        arrs = [np.ascontiguousarray(g["row_ptr"], np.int32),
                np.ascontiguousarray(g["ilabel"], np.int32),
                np.ascontiguousarray(g["olabel"], np.int32),
                np.ascontiguousarray(g["weight"], np.float32),
                np.ascontiguousarray(g["nextstate"], np.int32),
                np.ascontiguousarray(g["final"], np.float32)]
        _check(lib().kd_graph_create(device, g["num_states"], g["start"], *arrs,
                                     C.byref(self.h)))
        self.device = device

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.kd_graph_destroy(self.h)
            self.h = None

    @property
    def num_states(self):
        return lib().kd_graph_num_states(self.h)

    @property
    def num_arcs(self):
        return lib().kd_graph_num_arcs(self.h)

    @property
    def device_bytes(self):
        return lib().kd_graph_device_bytes(self.h)


class Path:
    def __init__(self, ok, il, ol, st, gc, ac, fw):
        self.ok = ok
        self.ilabel, self.olabel, self.state = il, ol, st
        self.graph_cost, self.ac_cost = gc, ac
        self.final_weight = fw

    @property
    def isyms(self):
        return self.ilabel[self.ilabel != 0]

    @property
    def osyms(self):
        return self.olabel[self.olabel != 0]

    @property
    def total_cost(self):
        return (float(self.graph_cost.astype(np.float64).sum() + self.final_weight[0]),
                float(self.ac_cost.astype(np.float64).sum() + self.final_weight[1]))


class RawLattice:
    """GetRawLattice output of one utterance as arrays (states are frame-major)."""

    def __init__(self, ok, num_states, start, src, dst, il, ol, gc, ac, sf, sg, fs, fg):
        self.ok, self.num_states, self.start = ok, num_states, start
        self.src, self.dst, self.ilabel, self.olabel = src, dst, il, ol
        self.graph_cost, self.ac_cost = gc, ac
        self.state_frame, self.state_graph = sf, sg
        self.final_state, self.final_graph = fs, fg

    def arc_set(self):
        f, s = self.state_frame, self.state_graph
        return set(zip(f[self.src].tolist(), s[self.src].tolist(), f[self.dst].tolist(),
                       s[self.dst].tolist(), self.ilabel.tolist(), self.olabel.tolist(),
                       self.graph_cost.tolist(), self.ac_cost.tolist()))

    def state_set(self):
        return set(zip(self.state_frame.tolist(), self.state_graph.tolist()))

    def final_set(self):
        return set(zip(self.state_graph[self.final_state].tolist(), self.final_graph.tolist()))


def default_lattice_options(**kw):
    o = LatticeOptions()
    lib().kd_lattice_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Batch:
    """`max_batch` independent FasterDecoder instances over one graph (kd_batch)."""

    def __init__(self, graph: Graph, max_batch=1, stream=None, **opts):
        self.graph = graph
        self.max_batch = max_batch
        self.opts = default_options(**opts)
        self.h = C.c_void_p()
        self.batch = 0
        _check(lib().kd_batch_create(graph.h, C.byref(self.opts), max_batch,
                                     C.c_void_p(stream or 0), C.byref(self.h)))

    def __del__(self):
        if getattr(self, "h", None) and _lib is not None:
            _lib.kd_batch_destroy(self.h)
            self.h = None

    def set_options(self, **opts):
        for k, v in opts.items():
            setattr(self.opts, k, v)
        _check(lib().kd_batch_set_options(self.h, C.byref(self.opts)))

    def set_log_softmax(self, on=True):
        """The matrices handed to advance / decode hold logits: log_softmax is fused into the
        kernel's row staging (kd_batch_set_log_softmax)."""
        _check(lib().kd_batch_set_log_softmax(self.h, 1 if on else 0))

    def init_decoding(self, batch=None):
        self.batch = self.max_batch if batch is None else batch
        _check(lib().kd_batch_init_decoding(self.h, self.batch))

    def _ptrs(self, ptrs):
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        return arr

    def advance_host(self, mats, offsets=None, max_num_frames=-1):
        """mats: list of C-contiguous float32 [T_i, V] arrays (host)."""
        assert len(mats) == self.batch
        mats = [np.ascontiguousarray(m, np.float32) for m in mats]
        V = mats[0].shape[1]
        rows = np.asarray([m.shape[0] for m in mats], np.int32)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
        _check(lib().kd_batch_advance(self.h, self._ptrs([m.ctypes.data for m in mats]), rows,
                                      None if off is None else off.ctypes.data, V, V,
                                      KD_MEM_HOST, max_num_frames))

    def advance_device(self, dev_ptrs, num_rows, num_cols, row_stride=None, offsets=None,
                       max_num_frames=-1, sync=True):
        """dev_ptrs: device addresses (ints) of row-major [T_i, V] float32 matrices."""
        rows = np.ascontiguousarray(num_rows, np.int32)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
        stride = num_cols if row_stride is None else row_stride
        fn = lib().kd_batch_advance if sync else lib().kd_batch_advance_async
        args = [self.h, self._ptrs(list(dev_ptrs)), rows,
                None if off is None else off.ctypes.data, num_cols, stride]
        if sync:
            args.append(KD_MEM_DEVICE)
        args.append(max_num_frames)
        _check(fn(*args))

    def decode_device(self, dev_ptrs, num_rows, num_cols, row_stride=None, offsets=None,
                      finalize=True, lattice=True, best_path=True, sync=True):
        """Decode() of every utterance in one launch (kd_batch_decode_async): InitDecoding,
        every frame and, as asked, FinalizeDecoding / GetRawLattice (lattice batches) or
        ReachedFinal / GetBestPath (1-best batches)."""
        rows = np.ascontiguousarray(num_rows, np.int32)
        self.batch = len(rows)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
        want = ((KD_DECODE_FINALIZE if finalize else 0) | (KD_DECODE_LATTICE if lattice else 0) |
                (KD_DECODE_BEST_PATH if best_path else 0))
        _check(lib().kd_batch_decode_async(self.h, self.batch, self._ptrs(list(dev_ptrs)), rows,
                                           None if off is None else off.ctypes.data, num_cols,
                                           num_cols if row_stride is None else row_stride, want))
        if sync:
            self.sync()

    def decode_tensor(self, log_probs, lengths=None, finalize=True, lattice=True, best_path=True,
                      sync=True):
        """Decode() of a whole padded batch held as ONE device tensor, the way icefall holds its
        `nnet_output`: `log_probs` float32 [B, T, V] on this batch's GPU (any batch / time
        strides, unit stride over V), `lengths` [B] valid frames per utterance (default: T).
        The per-utterance row-0 pointers are built vectorially from data_ptr() + i * stride:
        no per-utterance Python call.  Same launch as decode_device (kd_batch_decode_async;
        ref DecodableCtc(feats) + Decode(), csrc/decodable-ctc.cc:22-31)."""
        import torch
        t = log_probs
        if t.dim() != 3 or t.dtype != torch.float32 or not t.is_cuda:
            raise ValueError("decode_tensor needs a float32 CUDA tensor [B, T, V]")
        if t.stride(2) != 1:
            raise ValueError("decode_tensor needs unit stride over the last (V) dimension")
        B, T, V = t.shape
        if lengths is None:
            rows = np.full(B, T, np.int32)
        else:
            rows = np.ascontiguousarray(
                lengths.detach().cpu().numpy() if hasattr(lengths, "detach") else lengths, np.int32)
            if rows.shape != (B,) or (rows < 0).any() or (rows > T).any():
                raise ValueError("lengths must be [B] with 0 <= length <= T")
        ptrs = np.uint64(t.data_ptr()) + np.arange(B, dtype=np.uint64) * np.uint64(4 * t.stride(0))
        self.batch = B
        self._keep = t  # (the launch reads the tensor: it must outlive an asynchronous call)
        want = ((KD_DECODE_FINALIZE if finalize else 0) | (KD_DECODE_LATTICE if lattice else 0) |
                (KD_DECODE_BEST_PATH if best_path else 0))
        _check(lib().kd_batch_decode_async(self.h, B, ptrs.ctypes.data_as(C.POINTER(C.c_void_p)),
                                           rows, None, V, t.stride(1) if T > 1 else V, want))
        if sync:
            self.sync()

    def decode_host(self, mats, offsets=None, finalize=True, lattice=True, best_path=True, sync=True):
        """Decode() over host matrices (staged to the device first), one launch; with
        sync=False the call returns once the rows have LEFT the caller's memory -- pinned or
        pageable: the arrays may be refilled at once -- with the launch enqueued behind them
        (kd_batch_decode_host_async)."""
        mats = [np.ascontiguousarray(m, np.float32) for m in mats]
        self.batch = len(mats)
        V = mats[0].shape[1]
        rows = np.asarray([m.shape[0] for m in mats], np.int32)
        off = None if offsets is None else np.ascontiguousarray(offsets, np.int32)
        want = ((KD_DECODE_FINALIZE if finalize else 0) | (KD_DECODE_LATTICE if lattice else 0) |
                (KD_DECODE_BEST_PATH if best_path else 0))
        ptrs = np.asarray([m.ctypes.data for m in mats], np.uint64)
        self._keep = mats
        _check(lib().kd_batch_decode_host_async(self.h, self.batch,
                                                ptrs.ctypes.data_as(C.POINTER(C.c_void_p)), rows,
                                                None if off is None else off.ctypes.data, V, V, want))
        if sync:
            self.sync()

    def sync(self):
        _check(lib().kd_batch_sync(self.h))

    def capacity(self):
        out = np.zeros(8, np.int64)
        _check(lib().kd_batch_capacity(self.h, out))
        return dict(zip(CAPACITY_NAMES, out.tolist()))

    def split_info(self):
        """How calls of this batch are dealt out over the two kinds of workgroups (kd_hip.h)."""
        out = np.zeros(8, np.int32)
        _check(lib().kd_batch_split_info(self.h, out))
        return dict(zip(SPLIT_NAMES, out.tolist()))

    def set_lazy_prune(self, lazy):
        """PruneActiveTokens owed (run on un-finalized export or when a pool fills) or run when
        its frame comes, like the reference (kd_hip.h kd_batch_set_lazy_prune)."""
        _check(lib().kd_batch_set_lazy_prune(self.h, int(bool(lazy))))

    def sched_profile(self):
        """Cycles and counts per kind of workgroup (kd_hip.h kd_batch_sched_profile)."""
        out = np.zeros(16, np.int64)
        _check(lib().kd_batch_sched_profile(self.h, out))
        names = ("idle", "setup", "busy", "pop", "fresh", "resumed", "stolen", "workgroups")
        return {n: dict(small=int(out[2 * i]), large=int(out[2 * i + 1])) for i, n in enumerate(names)}

    def set_split(self, enable=-1, min_batch=-1, mig_lo=-1, mig_hi=-1, prune_small=-1, stay=-1):
        _check(lib().kd_batch_set_split(self.h, int(enable), min_batch, mig_lo, mig_hi, int(prune_small),
                                        int(stay)))

    def num_frames_decoded(self):
        out = np.zeros(self.batch, np.int32)
        _check(lib().kd_batch_num_frames_decoded(self.h, out))
        return out

    def status(self):
        out = np.zeros(self.batch, np.uint32)
        _check(lib().kd_batch_status(self.h, out))
        return out

    def reached_final(self):
        out = np.zeros(self.batch, np.int32)
        _check(lib().kd_batch_reached_final(self.h, out))
        return out.astype(bool)

    def best_paths(self, use_final_probs=True):
        lens = np.zeros(self.batch, np.int32)
        _check(lib().kd_batch_best_path_prepare(self.h, int(use_final_probs), lens))
        n = np.maximum(lens, 0).astype(np.int64)
        offs = np.zeros(self.batch, np.int64)
        offs[1:] = np.cumsum(n)[:-1]
        tot = max(int(n.sum()), 1)
        il, ol, st = (np.zeros(tot, np.int32) for _ in range(3))
        gc, ac = np.zeros(tot, np.float32), np.zeros(tot, np.float32)
        fw = np.zeros(2 * self.batch, np.float32)
        _check(lib().kd_batch_best_path_copy(self.h, offs, il, ol, st, gc, ac, fw))
        out = []
        for i in range(self.batch):
            a, b = int(offs[i]), int(offs[i] + n[i])
            out.append(Path(bool(lens[i] >= 0), il[a:b].copy(), ol[a:b].copy(), st[a:b].copy(),
                            gc[a:b].copy(), ac[a:b].copy(),
                            (float(fw[2 * i]), float(fw[2 * i + 1]))))
        return out

    def frame_tokens(self, utt, frame_plus_one):
        cnt = C.c_int32(0)
        _check(lib().kd_batch_frame_tokens(self.h, utt, frame_plus_one, 0, C.byref(cnt), None, None))
        n = cnt.value
        s = np.zeros(max(n, 1), np.int32)
        c = np.zeros(max(n, 1), np.float64)
        if n:
            _check(lib().kd_batch_frame_tokens(self.h, utt, frame_plus_one, n, C.byref(cnt),
                                               s.ctypes.data, c.ctypes.data))
        return s[:n], c[:n]

    def stats(self):
        out = np.zeros((self.batch, len(STAT_NAMES)), np.int64)
        _check(lib().kd_batch_stats(self.h, out))
        return out

    def phase_cycles(self):
        out = np.zeros((self.batch, 24), np.int64)
        _check(lib().kd_batch_phase_cycles(self.h, out))
        return out

    def last_kernel_ms(self):
        ms = C.c_float(0)
        _check(lib().kd_batch_last_kernel_ms(self.h, C.byref(ms)))
        return ms.value

    @property
    def device_bytes(self):
        return lib().kd_batch_device_bytes(self.h)


class LatticeBatch(Batch):
    """`max_batch` lattice decoders (LatticeSimpleDecoder semantics; with max_active /
    min_active the LatticeFasterDecoder configuration) over one graph."""

    def __init__(self, graph: Graph, max_batch=1, stream=None, **opts):
        self.graph = graph
        self.max_batch = max_batch
        self.lopts = default_lattice_options(**opts)
        self.h = C.c_void_p()
        self.batch = 0
        _check(lib().kd_batch_create_lattice(graph.h, C.byref(self.lopts), max_batch,
                                             C.c_void_p(stream or 0), C.byref(self.h)))

    def finalize_decoding(self):
        _check(lib().kd_batch_finalize_decoding(self.h))

    def final_relative_cost(self):
        out = np.zeros(self.batch, np.float32)
        _check(lib().kd_batch_final_relative_cost(self.h, out))
        return out

    def set_output_buffer(self, words):
        """Lends the batch a device buffer (a 1-D int64 CUDA tensor) for its exported lattices:
        the kernel writes them there instead of into the batch's pinned host pool, so that a
        gather can hand the buffer to RCCL as it is.  None: back to the host pool."""
        if words is None:
            self._out_buf = None
            _check(lib().kd_batch_set_output_buffer(self.h, None, 0))
            return
        import torch
        if not (words.is_cuda and words.dtype == torch.int64 and words.dim() == 1 and words.is_contiguous()):
            raise ValueError("set_output_buffer needs a contiguous 1-D int64 CUDA tensor")
        self._out_buf = words  # (kept alive: the kernel writes into it)
        _check(lib().kd_batch_set_output_buffer(self.h, words.data_ptr(), words.numel()))

    def raw_lattice_table(self):
        """After a decode that exported the lattices: (words in use, table [B, 5] of {first
        word, states, arcs, finals, start state}) -- with a lent output buffer `buf`,
        buf[:used] and the table are the call's whole result (sharded.unpack_pool reads them)."""
        used = C.c_int64(0)
        table = np.zeros((self.batch, 5), np.int64)
        _check(lib().kd_batch_raw_lattice_table(self.h, C.byref(used), table))
        return int(used.value), table

    def raw_lattices(self, use_final_probs=True):
        """GetRawLattice of every utterance: a sequence of RawLattice (built on access --
        a call that decodes a thousand utterances hands back two dozen arrays, not
        thirteen thousand views)."""
        sizes = np.zeros(4 * self.batch, np.int32)
        _check(lib().kd_batch_raw_lattice_prepare(self.h, int(use_final_probs), sizes))
        sz = sizes.reshape(self.batch, 4).astype(np.int64)
        so, ao, fo = (np.zeros(self.batch + 1, np.int64) for _ in range(3))
        so[1:], ao[1:], fo[1:] = np.cumsum(sz[:, 0]), np.cumsum(sz[:, 1]), np.cumsum(sz[:, 2])
        ns, na, nf = max(int(so[-1]), 1), max(int(ao[-1]), 1), max(int(fo[-1]), 1)
        src, dst, il, ol = (np.empty(na, np.int32) for _ in range(4))
        gc, ac = np.empty(na, np.float32), np.empty(na, np.float32)
        sf, sg = np.empty(ns, np.int32), np.empty(ns, np.int32)
        fs, fg = np.empty(nf, np.int32), np.empty(nf, np.float32)
        _check(lib().kd_batch_raw_lattice_copy(self.h, so[:-1].copy(), ao[:-1].copy(),
                                               fo[:-1].copy(), src, dst, il, ol, gc, ac, sf,
                                               sg, fs, fg))
        return RawLattices(sz, so, ao, fo, (src, dst, il, ol, gc, ac), (sf, sg), (fs, fg))


class RawLattices:
    """The raw lattices of a batch: batch-wide arrays plus per-utterance offsets; item i
    is a RawLattice of views into them."""

    def __init__(self, sizes, state_off, arc_off, final_off, arc_arrays, state_arrays, final_arrays):
        self.sizes, self.state_off, self.arc_off, self.final_off = sizes, state_off, arc_off, final_off
        self.arc_arrays, self.state_arrays, self.final_arrays = arc_arrays, state_arrays, final_arrays

    def __len__(self):
        return len(self.sizes)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        a, b = int(self.arc_off[i]), int(self.arc_off[i + 1])
        s0, s1 = int(self.state_off[i]), int(self.state_off[i + 1])
        f0, f1 = int(self.final_off[i]), int(self.final_off[i + 1])
        return RawLattice(bool(self.sizes[i, 0] > 0), int(self.sizes[i, 0]), int(self.sizes[i, 3]),
                          *(x[a:b] for x in self.arc_arrays), *(x[s0:s1] for x in self.state_arrays),
                          *(x[f0:f1] for x in self.final_arrays))

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def arcs_per_utt(self):
        return self.sizes[:, 1]

    @property
    def states_per_utt(self):
        return self.sizes[:, 0]


class SimpleBatch(Batch):
    """`max_batch` SimpleDecoder instances (plain beam) over one graph."""

    def __init__(self, graph: Graph, max_batch=1, beam=16.0, stream=None, arena_tokens=0,
                 table_slots=0, max_frames=0):
        self.graph = graph
        self.max_batch = max_batch
        self.h = C.c_void_p()
        self.batch = 0
        _check(lib().kd_batch_create_simple(graph.h, beam, arena_tokens, table_slots, max_frames,
                                            max_batch, C.c_void_p(stream or 0), C.byref(self.h)))

    def final_relative_cost(self):
        out = np.zeros(self.batch, np.float32)
        _check(lib().kd_batch_final_relative_cost(self.h, out))
        return out
