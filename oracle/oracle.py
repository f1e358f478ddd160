"""ctypes front end of the CPU oracle (oracle/kd_oracle.cc).

TEST INFRASTRUCTURE ONLY -- import it from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the kaldi_decoder package.
PARITY UNPINNED: see the header of kd_oracle.cc for what is and is not pinned
against the reference.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")

INT_MAX = 2**31 - 1


def build():
    """Compile the oracle (and the reference HashList pin when possible)."""
    subprocess.run(["make", "-C", _HERE, "--no-print-directory"], check=True,
                   stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libkd_oracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.kdo_last_error.restype = C.c_char_p
    L.kdo_graph_create.restype = C.c_void_p
    L.kdo_graph_create.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _f32p, _i32p, _f32p]
    L.kdo_graph_free.argtypes = [C.c_void_p]
    L.kdo_faster_create.restype = C.c_void_p
    L.kdo_faster_create.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_int32, C.c_float,
                                    C.c_float, C.c_int32]
    path_args = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _i32p, _i32p, _i32p,
                 _f32p, _f32p, _f32p]
    adv_args = [C.c_void_p, _f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    for pre in ("faster", "simple", "csimple"):
        getattr(L, f"kdo_{pre}_free").argtypes = [C.c_void_p]
        getattr(L, f"kdo_{pre}_init").argtypes = [C.c_void_p]
        getattr(L, f"kdo_{pre}_advance").argtypes = adv_args
        getattr(L, f"kdo_{pre}_num_frames_decoded").argtypes = [C.c_void_p]
        getattr(L, f"kdo_{pre}_reached_final").argtypes = [C.c_void_p]
        getattr(L, f"kdo_{pre}_num_tokens").argtypes = [C.c_void_p]
        getattr(L, f"kdo_{pre}_best_path").argtypes = path_args
    L.kdo_faster_tokens.argtypes = [C.c_void_p, _i32p, _f64p]
    L.kdo_faster_stats.argtypes = [C.c_void_p, _i64p]
    L.kdo_simple_create.restype = C.c_void_p
    L.kdo_simple_create.argtypes = [C.c_void_p, C.c_float]
    L.kdo_simple_final_relative_cost.restype = C.c_float
    L.kdo_simple_final_relative_cost.argtypes = [C.c_void_p]
    L.kdo_csimple_create.restype = C.c_void_p
    L.kdo_csimple_create.argtypes = [C.c_void_p, C.c_float]
    L.kdo_csimple_final_relative_cost.restype = C.c_float
    L.kdo_csimple_final_relative_cost.argtypes = [C.c_void_p]
    L.kdo_csimple_tokens.argtypes = [C.c_void_p, _i32p, _f64p]
    L.kdo_latsimple_create.restype = C.c_void_p
    L.kdo_latsimple_create.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_float]
    L.kdo_latsimple_free.argtypes = [C.c_void_p]
    L.kdo_latsimple_init.argtypes = [C.c_void_p]
    L.kdo_latsimple_decode.argtypes = [C.c_void_p, _f32p, C.c_int32, C.c_int32, C.c_int32]
    L.kdo_latsimple_decode_nofinal.argtypes = [C.c_void_p, _f32p, C.c_int32, C.c_int32, C.c_int32]
    L.kdo_latsimple_finalize.argtypes = [C.c_void_p]
    L.kdo_latsimple_num_frames_decoded.argtypes = [C.c_void_p]
    L.kdo_latsimple_final_relative_cost.restype = C.c_float
    L.kdo_latsimple_final_relative_cost.argtypes = [C.c_void_p]
    L.kdo_latsimple_links_created.restype = C.c_int64
    L.kdo_latsimple_links_created.argtypes = [C.c_void_p]
    L.kdo_latsimple_raw_lattice.argtypes = [C.c_void_p, C.c_int32, _i32p]
    L.kdo_latsimple_copy_lattice.argtypes = [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _f32p, _f32p,
                                             _i32p, _f32p, _i32p]
    L.kdo_latsimple_best_path.argtypes = path_args
    lat_copy = [C.c_void_p, _i32p, _i32p, _i32p, _i32p, _f32p, _f32p, _i32p, _f32p, _i32p, _i32p]
    L.kdo_latsimple_copy_lattice2.argtypes = lat_copy
    L.kdo_canonlat_create.restype = C.c_void_p
    L.kdo_canonlat_create.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_int32, C.c_float]
    L.kdo_canonlat_free.argtypes = [C.c_void_p]
    L.kdo_canonlat_decode.argtypes = [C.c_void_p, _f32p, C.c_int32, C.c_int32, C.c_int32]
    L.kdo_canonlat_num_frames_decoded.argtypes = [C.c_void_p]
    L.kdo_canonlat_final_relative_cost.restype = C.c_float
    L.kdo_canonlat_final_relative_cost.argtypes = [C.c_void_p]
    L.kdo_canonlat_links_created.restype = C.c_int64
    L.kdo_canonlat_links_created.argtypes = [C.c_void_p]
    L.kdo_canonlat_frame_tokens.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, _f32p, _f32p]
    L.kdo_canonlat_raw_lattice.argtypes = [C.c_void_p, _i32p]
    L.kdo_canonlat_created_links.restype = C.c_int64
    L.kdo_canonlat_created_links.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    L.kdo_canonlat_copy_lattice.argtypes = lat_copy
    L.kdo_canonlat_best_path.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), _i32p, _i32p,
                                         _i32p, _f32p, _f32p, _f32p]
    _bind_hashlist(L, "kdo_hashlist")
    _LIB = L
    return L


def _bind_hashlist(L, pre):
    getattr(L, f"{pre}_create").restype = C.c_void_p
    getattr(L, f"{pre}_free").argtypes = [C.c_void_p]
    getattr(L, f"{pre}_set_size").argtypes = [C.c_void_p, C.c_int64]
    getattr(L, f"{pre}_insert").argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    getattr(L, f"{pre}_find").argtypes = [C.c_void_p, C.c_int32]
    getattr(L, f"{pre}_clear").argtypes = [C.c_void_p, C.c_int32, _i32p, _i32p]
    getattr(L, f"{pre}_list").argtypes = [C.c_void_p, C.c_int32, _i32p, _i32p]


def ref_lib():
    """The reference's own HashList (oracle/_ref/), or None if it was never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libref_hashlist.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
        _bind_hashlist(_REF, "ref_hashlist")
    return _REF


def _err():
    return RuntimeError(lib().kdo_last_error().decode())


class HashList:
    """Either the oracle's restatement (ref=False) or the reference's own (ref=True)."""

    def __init__(self, ref=False):
        self._L = ref_lib() if ref else lib()
        self._p = "ref_hashlist" if ref else "kdo_hashlist"
        self._h = C.c_void_p(getattr(self._L, f"{self._p}_create")())

    def __del__(self):
        getattr(self._L, f"{self._p}_free")(self._h)

    def set_size(self, n):
        getattr(self._L, f"{self._p}_set_size")(self._h, n)

    def insert(self, k, v):
        return getattr(self._L, f"{self._p}_insert")(self._h, k, v)

    def find(self, k):
        return getattr(self._L, f"{self._p}_find")(self._h, k)

    def _dump(self, fn, cap=1 << 16):
        k = np.zeros(cap, np.int32)
        v = np.zeros(cap, np.int32)
        n = getattr(self._L, f"{self._p}_{fn}")(self._h, cap, k, v)
        assert n <= cap
        return list(zip(k[:n].tolist(), v[:n].tolist()))

    def clear(self):
        return self._dump("clear")

    def list(self):
        return self._dump("list")


class Graph:
    def __init__(self, g):
        L = lib()
        self.g = g
        self._keep = [np.ascontiguousarray(g[k]) for k in
                      ("row_ptr", "ilabel", "olabel", "weight", "nextstate", "final")]
        self.h = C.c_void_p(L.kdo_graph_create(g["num_states"], g["start"], *self._keep))

    def __del__(self):
        if _LIB is not None:  # (None once the interpreter is shutting down)
            _LIB.kdo_graph_free(self.h)


class BestPath:
    def __init__(self, ok, il, ol, st, gc, ac, fw):
        self.ok = ok
        self.ilabel, self.olabel, self.state = il, ol, st
        self.graph_cost, self.ac_cost = gc, ac
        self.final_weight = fw  # (graph, acoustic)

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


def _best_path(fn, h, use_final_probs, cap=1 << 16):
    n = C.c_int32(0)
    il = np.zeros(cap, np.int32)
    ol = np.zeros(cap, np.int32)
    st = np.zeros(cap, np.int32)
    gc = np.zeros(cap, np.float32)
    ac = np.zeros(cap, np.float32)
    fw = np.zeros(2, np.float32)
    r = fn(h, int(use_final_probs), cap, C.byref(n), il, ol, st, gc, ac, fw)
    if r < 0:
        raise _err()
    k = n.value
    return BestPath(bool(r), il[:k].copy(), ol[:k].copy(), st[:k].copy(), gc[:k].copy(),
                    ac[:k].copy(), (float(fw[0]), float(fw[1])))


class _Decoder:
    _pre = ""

    h = None

    def __del__(self):
        if self.h and _LIB is not None:
            getattr(_LIB, f"kdo_{self._pre}_free")(self.h)

    def init_decoding(self):
        if getattr(lib(), f"kdo_{self._pre}_init")(self.h):
            raise _err()

    def advance_decoding(self, log_probs, offset=0, max_num_frames=-1):
        lp = np.ascontiguousarray(log_probs, dtype=np.float32)
        if getattr(lib(), f"kdo_{self._pre}_advance")(self.h, lp, lp.shape[0], lp.shape[1],
                                                      offset, max_num_frames):
            raise _err()

    def decode(self, log_probs):
        self.init_decoding()
        self.advance_decoding(log_probs)

    def num_frames_decoded(self):
        return getattr(lib(), f"kdo_{self._pre}_num_frames_decoded")(self.h)

    def reached_final(self):
        return bool(getattr(lib(), f"kdo_{self._pre}_reached_final")(self.h))

    def num_tokens(self):
        return getattr(lib(), f"kdo_{self._pre}_num_tokens")(self.h)

    def get_best_path(self, use_final_probs=True):
        return _best_path(getattr(lib(), f"kdo_{self._pre}_best_path"), self.h, use_final_probs)


class FasterDecoder(_Decoder):
    """policy 0 = reference order, policy 1 = canonical (what the GPU computes)."""
    _pre = "faster"

    def __init__(self, graph: Graph, beam=16.0, max_active=INT_MAX, min_active=20,
                 beam_delta=0.5, hash_ratio=2.0, policy=0):
        self.graph = graph
        h = lib().kdo_faster_create(graph.h, beam, max_active, min_active, beam_delta,
                                    hash_ratio, policy)
        if not h:
            raise _err()
        self.h = C.c_void_p(h)

    def tokens(self):
        n = self.num_tokens()
        s = np.zeros(n, np.int32)
        c = np.zeros(n, np.float64)
        lib().kdo_faster_tokens(self.h, s, c)
        return s, c

    def stats(self):
        v = np.zeros(9, np.int64)
        lib().kdo_faster_stats(self.h, v)
        keys = ("frames", "tokens_in", "tokens_expanded", "arcs_scanned", "arcs_emit_kept",
                "tokens_out", "eps_arcs_scanned", "eps_states_expanded", "max_tokens")
        return dict(zip(keys, v.tolist()))


class SimpleDecoder(_Decoder):
    _pre = "simple"

    def __init__(self, graph: Graph, beam=16.0):
        self.graph = graph
        self.h = C.c_void_p(lib().kdo_simple_create(graph.h, beam))

    def final_relative_cost(self):
        return lib().kdo_simple_final_relative_cost(self.h)


class CanonSimpleDecoder(_Decoder):
    """Order-free SimpleDecoder (policy 1): what the GPU computes for SimpleDecoder."""
    _pre = "csimple"

    def __init__(self, graph: Graph, beam=16.0):
        self.graph = graph
        self.h = C.c_void_p(lib().kdo_csimple_create(graph.h, beam))

    def final_relative_cost(self):
        return lib().kdo_csimple_final_relative_cost(self.h)

    def tokens(self):
        n = self.num_tokens()
        s = np.zeros(n, np.int32)
        c = np.zeros(n, np.float64)
        lib().kdo_csimple_tokens(self.h, s, c)
        return s, c


class RawLattice:
    def __init__(self, ok, num_states, start, src, dst, il, ol, gc, ac, fs, fg, frame, gstate):
        self.ok, self.num_states, self.start = ok, num_states, start
        self.src, self.dst, self.ilabel, self.olabel = src, dst, il, ol
        self.graph_cost, self.ac_cost = gc, ac
        self.final_state, self.final_graph = fs, fg
        self.state_frame, self.state_graph = frame, gstate

    def arc_set(self):
        """Arcs as tuples keyed by (frame, graph state) of both ends: a numbering-free
        form of the lattice for comparisons."""
        f, s = self.state_frame, self.state_graph
        return set(zip(f[self.src].tolist(), s[self.src].tolist(), f[self.dst].tolist(),
                       s[self.dst].tolist(), self.ilabel.tolist(), self.olabel.tolist(),
                       self.graph_cost.tolist(), self.ac_cost.tolist()))

    def state_set(self):
        return set(zip(self.state_frame.tolist(), self.state_graph.tolist()))

    def final_set(self):
        return set(zip(self.state_graph[self.final_state].tolist(), self.final_graph.tolist()))


def _fetch_lattice(ok, sizes, copy_fn, h):
    ns, na, nf, start = sizes.tolist()
    src, dst, il, ol = (np.zeros(na, np.int32) for _ in range(4))
    gc, ac = np.zeros(na, np.float32), np.zeros(na, np.float32)
    fs, fg = np.zeros(nf, np.int32), np.zeros(nf, np.float32)
    frame, gstate = np.zeros(ns, np.int32), np.zeros(ns, np.int32)
    copy_fn(h, src, dst, il, ol, gc, ac, fs, fg, frame, gstate)
    return RawLattice(ok, ns, start, src, dst, il, ol, gc, ac, fs, fg, frame, gstate)


class LatticeSimpleDecoder:
    h = None

    def __init__(self, graph: Graph, beam=16.0, lattice_beam=10.0, prune_interval=25,
                 prune_scale=0.1):
        self.graph = graph
        h = lib().kdo_latsimple_create(graph.h, beam, lattice_beam, prune_interval, prune_scale)
        if not h:
            raise _err()
        self.h = C.c_void_p(h)

    def __del__(self):
        if self.h and _LIB is not None:
            _LIB.kdo_latsimple_free(self.h)

    def decode(self, log_probs, offset=0):
        lp = np.ascontiguousarray(log_probs, dtype=np.float32)
        r = lib().kdo_latsimple_decode(self.h, lp, lp.shape[0], lp.shape[1], offset)
        if r < 0:
            raise _err()
        return bool(r)

    def decode_nofinal(self, log_probs, offset=0):
        """Decode()'s loop without the FinalizeDecoding that ends it (lattice-simple-decoder.cc:53-66)."""
        lp = np.ascontiguousarray(log_probs, dtype=np.float32)
        if lib().kdo_latsimple_decode_nofinal(self.h, lp, lp.shape[0], lp.shape[1], offset) != 0:
            raise _err()

    def finalize_decoding(self):
        if lib().kdo_latsimple_finalize(self.h) != 0:
            raise _err()

    def num_frames_decoded(self):
        return lib().kdo_latsimple_num_frames_decoded(self.h)

    def final_relative_cost(self):
        return lib().kdo_latsimple_final_relative_cost(self.h)

    def links_created(self):
        return lib().kdo_latsimple_links_created(self.h)

    def get_raw_lattice(self, use_final_probs=True):
        sizes = np.zeros(4, np.int32)
        r = lib().kdo_latsimple_raw_lattice(self.h, int(use_final_probs), sizes)
        if r < 0:
            raise _err()
        return _fetch_lattice(bool(r), sizes, lib().kdo_latsimple_copy_lattice2, self.h)

    def get_best_path(self, use_final_probs=True):
        return _best_path(lib().kdo_latsimple_best_path, self.h, use_final_probs)


class CanonLatticeDecoder:
    """Order-free lattice decoder (policy 1): what the GPU lattice decoders compute.
    max_active / min_active at their defaults == LatticeSimpleDecoder semantics."""
    h = None

    def __init__(self, graph: Graph, beam=16.0, lattice_beam=10.0, max_active=INT_MAX,
                 min_active=0, beam_delta=0.5):
        self.graph = graph
        h = lib().kdo_canonlat_create(graph.h, beam, lattice_beam, max_active, min_active,
                                      beam_delta)
        if not h:
            raise _err()
        self.h = C.c_void_p(h)

    def __del__(self):
        if self.h and _LIB is not None:
            _LIB.kdo_canonlat_free(self.h)

    def decode(self, log_probs, offset=0):
        lp = np.ascontiguousarray(log_probs, dtype=np.float32)
        r = lib().kdo_canonlat_decode(self.h, lp, lp.shape[0], lp.shape[1], offset)
        if r < 0:
            raise _err()
        return bool(r)

    def num_frames_decoded(self):
        return lib().kdo_canonlat_num_frames_decoded(self.h)

    def final_relative_cost(self):
        return lib().kdo_canonlat_final_relative_cost(self.h)

    def links_created(self):
        return lib().kdo_canonlat_links_created(self.h)

    def frame_tokens(self, f, cap=1 << 20):
        st = np.zeros(cap, np.int32)
        tot = np.zeros(cap, np.float32)
        ex = np.zeros(cap, np.float32)
        n = lib().kdo_canonlat_frame_tokens(self.h, f, cap, st, tot, ex)
        return st[:n].copy(), tot[:n].copy(), ex[:n].copy()

    def created_links(self):
        """Every forward link created (before any pruning) as rows of
        (src frame, src state, dst frame, dst state, ilabel, olabel, graph, ac)."""
        n = lib().kdo_canonlat_created_links(self.h, 0, None)
        rows = np.zeros((n, 8), np.float64)
        lib().kdo_canonlat_created_links(self.h, n, rows.ctypes.data)
        return rows

    def get_raw_lattice(self):
        sizes = np.zeros(4, np.int32)
        r = lib().kdo_canonlat_raw_lattice(self.h, sizes)
        if r < 0:
            raise _err()
        return _fetch_lattice(bool(r), sizes, lib().kdo_canonlat_copy_lattice, self.h)

    def get_best_path(self):
        cap = 1 << 16
        n = C.c_int32(0)
        il, ol, st = (np.zeros(cap, np.int32) for _ in range(3))
        gc, ac = np.zeros(cap, np.float32), np.zeros(cap, np.float32)
        fw = np.zeros(2, np.float32)
        r = lib().kdo_canonlat_best_path(self.h, cap, C.byref(n), il, ol, st, gc, ac, fw)
        if r < 0:
            raise _err()
        k = n.value
        return BestPath(bool(r), il[:k].copy(), ol[:k].copy(), st[:k].copy(), gc[:k].copy(),
                        ac[:k].copy(), (float(fw[0]), float(fw[1])))
