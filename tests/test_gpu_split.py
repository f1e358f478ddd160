"""Two kinds of workgroups in one call (round 6): an utterance's frames may be run by "large"
workgroups (one per CU) or "small" ones (three per CU), changing hands at frame boundaries.
The reference runs every frame of an utterance in one thread (lattice-simple-decoder.cc:53-66,
faster-decoder.cc:126-152): WHICH workgroup runs a frame must change nothing -- lattices, paths
and exported arrays are compared with the canonical oracle and with the same decode by one
kernel, at thresholds that make utterances change hands every few frames, at every frame that
grows or shrinks past a handful of tokens, only at prune points, and never."""
import numpy as np
import pytest

from kaldi_decoder import cabi, synth

pytestmark = pytest.mark.gpu

HLG = dict(V=100, n_words=2000, n_hist=2000, avg_succ=10, seed=2)


def _need_split(bt):
    """The batch prepared for split launches (it is when first asked), or the test skipped."""
    try:
        bt.set_split(enable=1)
    except RuntimeError as e:
        if "second kernel" in str(e):
            pytest.skip("no second kernel on this box (CU-masked streams unavailable)")
        raise
    return bt.split_info()


def _lattice_arrays(lat):
    return [np.asarray(getattr(lat, k)) for k in ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost", "state_frame", "state_graph", "final_state", "final_graph")]


SETTINGS = [
    dict(mig_lo=3, mig_hi=6),                    # hand-overs at a handful of tokens
    dict(mig_lo=40, mig_hi=40),
    dict(mig_lo=0, mig_hi=256, prune_small=1),   # the small kind prunes, the large kind searches ... mostly
    dict(mig_lo=256, mig_hi=256, prune_small=1),
    dict(mig_lo=1, mig_hi=1),                    # everything but single-token frontiers on the large kind
    dict(mig_lo=3, mig_hi=6, stay=0),            # surplus workgroups leave the call early
]


@pytest.mark.parametrize("setting", SETTINGS, ids=[str(sorted(s.items())) for s in SETTINGS])
def test_lattices_do_not_depend_on_who_ran_a_frame(oracle, setting):
    import torch
    g = synth.hlg_graph(**HLG)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 48, 120, HLG["V"]
    lps = [synth.make_logprobs(T, V, 300 + i, peak=8.0 if i % 3 else 3.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    opts = dict(beam=12.0, lattice_beam=6.0, min_active=0, prune_interval=10)
    one = cabi.LatticeBatch(G, max_batch=B, **opts)
    one.set_split(enable=0)
    one.decode_tensor(dev)
    base = one.raw_lattices()
    assert one.stats()[:, cabi.STAT_NAMES.index("handovers")].sum() == 0
    two = cabi.LatticeBatch(G, max_batch=B, **opts)
    two.set_lazy_prune(0)  # (prunes run when their frames come: `prune_small` has something to hand over)
    _need_split(two)
    # (stay: 48 utterances on a chip of hundreds of workgroups -- left to themselves the surplus
    # workgroups go home at once, of either kind, and nothing is left to change hands with)
    two.set_split(enable=1, min_batch=1, prune_small=setting.get("prune_small", 0),
                  mig_lo=setting["mig_lo"], mig_hi=setting["mig_hi"], stay=setting.get("stay", 1))
    for rep in range(2):  # (a batch object's second call starts from the scheduler state the first left)
        two.decode_tensor(dev)
        assert two.num_frames_decoded().tolist() == [T] * B and not two.status().any()
        lats = two.raw_lattices()
        for i in range(B):
            for a, b in zip(_lattice_arrays(base[i]), _lattice_arrays(lats[i])):
                assert a.tobytes() == b.tobytes(), (setting, rep, i)
        assert np.array_equal(one.final_relative_cost(), two.final_relative_cost())
    st = dict(zip(cabi.STAT_NAMES, two.stats().sum(axis=0).tolist()))
    so = dict(zip(cabi.STAT_NAMES, one.stats().sum(axis=0).tolist()))
    for k in ("frames", "tokens_in", "tokens_expanded", "arcs_scanned", "tokens_out", "links",
              "eps_arcs_scanned", "eps_expansions"):
        assert st[k] == so[k], k  # the order-free work counters: the same decode
    # (which kind ran how many frames depends on the launch: a call of fewer utterances than the
    # chip has workgroups sends the surplus home early, of either kind)
    if setting.get("stay", 1):
        assert st["handovers"] > 0 and 0 < st["small_wg_frames"] < st["frames"]
    for i in (0, 1, 2, 17, 47):
        can = oracle.CanonLatticeDecoder(OG, beam=12.0, lattice_beam=6.0, min_active=0)
        can.decode(lps[i])
        cl = can.get_raw_lattice()
        assert lats[i].arc_set() == cl.arc_set() and lats[i].state_set() == cl.state_set(), i


def test_best_paths_do_not_depend_on_who_ran_a_frame(oracle):
    import torch
    g = synth.hl_graph(V=60, n_words=1500, seed=5)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 40, 150, 60
    lps = [synth.make_logprobs(T, V, 500 + i, peak=6.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    bt = cabi.Batch(G, max_batch=B, beam=10.0, max_active=300, min_active=20)
    _need_split(bt)
    bt.set_split(enable=1, min_batch=1, mig_lo=8, mig_hi=24, stay=1)
    bt.decode_tensor(dev)
    paths = bt.best_paths()
    assert bt.stats()[:, cabi.STAT_NAMES.index("handovers")].sum() > 0
    for i in range(0, B, 5):
        d = oracle.FasterDecoder(OG, beam=10.0, max_active=300, min_active=20, policy=1)
        d.decode(lps[i])
        p = d.get_best_path()
        for k in ("ilabel", "olabel", "state", "graph_cost", "ac_cost"):
            assert getattr(paths[i], k).tolist() == getattr(p, k).tolist(), (i, k)


def test_streaming_and_growing_while_utterances_change_hands(oracle):
    """AdvanceDecoding in chunks (each call a split launch that ends with utterances parked at
    frame boundaries) from pools that are too small (utterances stop mid-launch, the host grows
    the pools and launches again): the hand-over state and the growth path meet."""
    g = synth.hlg_graph(**HLG)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 12, 90, HLG["V"]
    lps = [synth.make_logprobs(T, V, 900 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(B)]
    lb = cabi.LatticeBatch(G, max_batch=B, beam=12.0, lattice_beam=6.0, min_active=0, prune_interval=7,
                           arena_tokens=1 << 12, arena_links=1 << 13, table_slots=1 << 14)
    _need_split(lb)
    lb.set_lazy_prune(0)
    lb.set_split(enable=1, min_batch=1, mig_lo=5, mig_hi=12, prune_small=1, stay=1)
    lb.init_decoding()
    for f0 in range(0, T, 13):
        lb.advance_host(lps, max_num_frames=13)
    assert lb.num_frames_decoded().tolist() == [T] * B
    assert lb.capacity()["grow_events"] > 0
    lb.finalize_decoding()
    lats = lb.raw_lattices()
    for i in range(B):
        can = oracle.CanonLatticeDecoder(OG, beam=12.0, lattice_beam=6.0, min_active=0)
        can.decode(lps[i])
        cl = can.get_raw_lattice()
        assert lats[i].arc_set() == cl.arc_set() and lats[i].state_set() == cl.state_set(), i


def test_split_settings_are_validated():
    g = synth.hlg_graph(**HLG)
    G = cabi.Graph(g)
    lb = cabi.LatticeBatch(G, max_batch=4, beam=12.0, lattice_beam=6.0)
    info = _need_split(lb)
    assert info["split"] == 1 and info["large_workgroups"] >= 1 and info["small_workgroups"] >= 1
    assert 0 <= info["mig_lo"] <= info["mig_hi"]
    with pytest.raises(RuntimeError, match="mig_lo"):
        lb.set_split(mig_lo=10, mig_hi=5)
    lb.set_split(enable=0)
    assert lb.split_info()["split"] == 0
    import os
    if os.environ.get("KD_SPLIT", "0") == "0" and os.environ.get("KD_HIP_LIB", "libkd_hip.so") == "libkd_hip.so":
        fresh = cabi.LatticeBatch(G, max_batch=4, beam=12.0, lattice_beam=6.0)
        assert fresh.split_info()["split"] == 0  # (calls are dealt out over two kernels only when asked)
    sb = cabi.SimpleBatch(G, max_batch=2, beam=10.0)
    assert sb.split_info()["large_workgroups"] == 0  # (SimpleDecoder: one kernel)
    with pytest.raises(RuntimeError, match="second kernel"):
        sb.set_split(enable=1)
