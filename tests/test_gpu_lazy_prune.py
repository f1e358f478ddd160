"""PruneActiveTokens owed or run (round 6; include/kd_hip.h kd_batch_set_lazy_prune).

The reference prunes every `prune_interval` frames (lattice-simple-decoder.cc:57-59).  The GPU
decoders may only OWE a prune when its frame comes and run it later -- the owed ones in order,
each exactly as it would have run then -- when something can tell the difference: GetRawLattice
of the un-finalized lattice, or a frame that finds no room in a pool.  Nothing observable may
depend on which it was: un-finalized lattices at several chunkings and intervals (the arrays,
byte for byte), finalized ones, FinalRelativeCost; and a long utterance in pools that may not
grow must still decode (the owed prunes are its garbage collection)."""
import numpy as np
import pytest

from kaldi_decoder import cabi, synth

pytestmark = pytest.mark.gpu
PRUNES = cabi.STAT_NAMES.index("prunes")


def _arrays(lat):
    return [np.asarray(getattr(lat, k)).tobytes() for k in
            ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost", "state_frame", "state_graph",
             "final_state", "final_graph")]


@pytest.mark.parametrize("interval,chunk", [(5, 1), (5, 5), (5, 7), (10, 10), (3, 50), (25, 13)])
def test_unfinalized_lattices_do_not_depend_on_when_a_prune_ran(interval, chunk):
    g = synth.hlg_graph(V=60, n_words=600, n_hist=600, avg_succ=6, seed=9)
    G = cabi.Graph(g)
    B, T, V = 6, 64, 60
    lps = [synth.make_logprobs(T, V, 40 + i, peak=4.0, lexicon=g["lexicon"]) for i in range(B)]
    opts = dict(beam=11.0, lattice_beam=4.0, min_active=0, prune_interval=interval, prune_scale=0.1)
    seen = {}
    for lazy in (0, 1):
        lb = cabi.LatticeBatch(G, max_batch=B, **opts)
        lb.set_lazy_prune(lazy)
        lb.init_decoding()
        snaps = []
        for f0 in range(0, T, chunk):
            lb.advance_host(lps, max_num_frames=chunk)
            if (f0 // chunk) % 3 == 0 or f0 + chunk >= T:
                # GetRawLattice before FinalizeDecoding: what the prunes run so far have left
                snaps.append([_arrays(lat) for lat in lb.raw_lattices()])
        n_prunes = int(lb.stats()[:, PRUNES].sum())
        lb.finalize_decoding()
        snaps.append([_arrays(lat) for lat in lb.raw_lattices()])
        seen[lazy] = (snaps, lb.final_relative_cost().tolist(), n_prunes)
    assert seen[0][0] == seen[1][0]
    assert seen[0][1] == seen[1][1]
    # both ran every prune the reference has run by the last export before FinalizeDecoding
    assert seen[0][2] == seen[1][2] == B * ((T - 1) // interval)


def test_a_plain_decode_owes_its_prunes_and_never_pays(oracle):
    g = synth.hlg_graph(V=60, n_words=600, n_hist=600, avg_succ=6, seed=9)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 8, 100, 60
    lps = [synth.make_logprobs(T, V, 70 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(B)]
    opts = dict(beam=11.0, lattice_beam=5.0, min_active=0, prune_interval=10)
    out = {}
    for lazy in (0, 1):
        lb = cabi.LatticeBatch(G, max_batch=B, **opts)
        lb.set_lazy_prune(lazy)
        lb.decode_host(lps)  # InitDecoding .. FinalizeDecoding + GetRawLattice in one launch
        out[lazy] = ([_arrays(lat) for lat in lb.raw_lattices()], int(lb.stats()[:, PRUNES].sum()))
    assert out[0][0] == out[1][0]
    assert out[0][1] == B * ((T - 1) // 10) and out[1][1] == 0
    lats = cabi.LatticeBatch(G, max_batch=B, **opts)
    lats.set_lazy_prune(1)
    lats.decode_host(lps)
    for i in (0, 3, 7):
        can = oracle.CanonLatticeDecoder(OG, beam=11.0, lattice_beam=5.0, min_active=0)
        can.decode(lps[i])
        cl = can.get_raw_lattice()
        gl = lats.raw_lattices()[i]
        assert gl.arc_set() == cl.arc_set() and gl.state_set() == cl.state_set()


def test_owed_prunes_are_the_garbage_collection_of_pools_that_may_not_grow(oracle):
    """A 1500-frame utterance in pools of 2^16 tokens / 2^18 links with
    `no_grow`: without any pruning the pools overflow after a fraction of the frames; the owed
    prunes are run when a frame finds no room, the collector returns the old chunks, and the
    frame -- which was not committed -- starts again."""
    g = synth.ctc_topo(100)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    T, V = 1500, 100
    lp = synth.make_logprobs(T, V, 77, peak=7.0)
    opts = dict(beam=8.0, lattice_beam=4.0, min_active=0)
    kw = dict(arena_tokens=1 << 16, arena_links=1 << 18, max_frames=T, no_grow=1)
    off = cabi.LatticeBatch(G, max_batch=1, prune_interval=10**6, **kw, **opts)
    with pytest.raises(RuntimeError, match="pool full"):
        off.decode_host([lp])
    lb = cabi.LatticeBatch(G, max_batch=1, prune_interval=25, **kw, **opts)
    lb.set_lazy_prune(1)
    lb.decode_host([lp])
    st = dict(zip(cabi.STAT_NAMES, lb.stats()[0].tolist()))
    assert st["frames"] == T and 0 < st["prunes"] <= (T - 1) // 25
    gl = lb.raw_lattices()[0]
    can = oracle.CanonLatticeDecoder(OG, **opts)
    can.decode(lp)
    cl = can.get_raw_lattice()
    assert gl.arc_set() == cl.arc_set() and gl.state_set() == cl.state_set()
    assert st["tokens_out"] > (1 << 16)  # (more tokens than the pool holds went through it)
    assert lb.capacity()["grow_events"] == 0


def test_a_pool_at_its_budget_is_collected_not_grown(oracle, monkeypatch):
    """Pools may grow here, but the batch's budget (KD_GC_TOKENS / KD_GC_LINKS records; default
    2^29 / 2^30) is what they start with: a pool at its budget is garbage-collected by the owed
    prunes instead of grown.  Same decode as above, same lattice, prunes paid; without the budget
    the same pools simply grow and no prune is ever paid."""
    g = synth.ctc_topo(100)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    T, V = 1500, 100
    lp = synth.make_logprobs(T, V, 77, peak=7.0)
    opts = dict(beam=8.0, lattice_beam=4.0, min_active=0)
    kw = dict(arena_tokens=1 << 16, arena_links=1 << 18, max_frames=T)
    can = oracle.CanonLatticeDecoder(OG, **opts)
    can.decode(lp)
    cl = can.get_raw_lattice()
    grown = cabi.LatticeBatch(G, max_batch=1, prune_interval=25, **kw, **opts)
    grown.set_lazy_prune(1)
    grown.decode_host([lp])
    assert int(grown.stats()[0, PRUNES]) == 0 and grown.capacity()["grow_events"] > 0
    monkeypatch.setenv("KD_GC_TOKENS", str(1 << 16))
    monkeypatch.setenv("KD_GC_LINKS", str(1 << 18))
    kept = cabi.LatticeBatch(G, max_batch=1, prune_interval=25, **kw, **opts)
    kept.set_lazy_prune(1)
    kept.decode_host([lp])
    assert int(kept.stats()[0, PRUNES]) > 0 and kept.capacity()["grow_events"] == 0
    for lb in (grown, kept):
        gl = lb.raw_lattices()[0]
        assert gl.arc_set() == cl.arc_set() and gl.state_set() == cl.state_set()
