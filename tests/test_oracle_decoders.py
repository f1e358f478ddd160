"""Oracle self-consistency and known-answer tests (CPU only).

The reference ships no decoder tests, so the restatement in
oracle/kd_oracle.cc is pinned by (a) an independent exact Viterbi
(tests/viterbi_ref.py), (b) the cross-decoder invariants the reference satisfies
by construction (FasterDecoder == SimpleDecoder == shortest path through the
LatticeSimpleDecoder lattice), (c) policy 0 (reference order) vs policy 1
(canonical, what the GPU computes).
"""
import numpy as np
import pytest

from kaldi_decoder import synth
from viterbi_ref import viterbi

BIG = 1.0e10


def _graphs():
    yield "H8", synth.ctc_topo(8), None
    yield "H32", synth.ctc_topo(32), None
    for seed in range(4):
        yield f"rand{seed}", synth.random_graph(60 + 40 * seed, 3.0, 12, 0.25, seed), None
    g = synth.hl_graph(V=20, n_words=60, seed=3)
    yield "HL", g, g["lexicon"]
    g = synth.hlg_graph(V=20, n_words=40, n_hist=40, avg_succ=5, seed=4)
    yield "HLG", g, g["lexicon"]


def _V(g):
    return int(g["ilabel"].max())


@pytest.mark.parametrize("policy", [0, 1])
def test_infinite_beam_equals_exact_viterbi(oracle, policy):
    for name, g, lex in _graphs():
        G = oracle.Graph(g)
        for utt in range(3):
            lp = synth.make_logprobs(25, _V(g), utt, peak=4.0, lexicon=lex)
            ref = viterbi(g, lp)
            d = oracle.FasterDecoder(G, beam=BIG, min_active=0, policy=policy)
            d.decode(lp)
            bp = d.get_best_path()
            assert bp.ok == ref["ok"], name
            assert d.reached_final() == ref["reached_final"], name
            assert bp.ilabel.tolist() == ref["ilabel"].tolist(), (name, utt)
            assert bp.olabel.tolist() == ref["olabel"].tolist(), (name, utt)
            assert bp.state.tolist() == ref["state"].tolist(), (name, utt)
            tot = sum(bp.total_cost)
            assert abs(tot - ref["total"]) <= 1e-4 * max(1.0, abs(ref["total"])), (name, utt)
            # frontier costs are exact doubles
            s, c = d.tokens()
            assert np.isclose(c.min(), ref["cost"] if not ref["reached_final"] else c.min())


@pytest.mark.parametrize("beam", [4.0, 8.0, 15.0])
def test_policies_agree_without_active_constraints(oracle, beam):
    # max_active=INT_MAX, min_active=0: the in-beam token set is order free
    for name, g, lex in _graphs():
        G = oracle.Graph(g)
        lp = synth.make_logprobs(40, _V(g), 7, peak=6.0, lexicon=lex)
        d0 = oracle.FasterDecoder(G, beam=beam, min_active=0, policy=0)
        d1 = oracle.FasterDecoder(G, beam=beam, min_active=0, policy=1)
        d0.init_decoding()
        d1.init_decoding()
        for t in range(lp.shape[0]):
            d0.advance_decoding(lp, max_num_frames=1)
            d1.advance_decoding(lp, max_num_frames=1)
            s0, c0 = d0.tokens()
            s1, c1 = d1.tokens()
            # canonical tokens are a subset of the reference-order tokens, equal costs
            pos = np.searchsorted(s0, s1)
            assert (pos < len(s0)).all() and (s0[pos] == s1).all(), (name, t)
            assert (c0[pos] == c1).all(), (name, t)
            # and every reference-only token lies at or beyond the canonical frontier's beam
            extra = np.setdiff1d(np.arange(len(s0)), pos)
            if len(extra) and len(c1):
                assert c0[extra].min() >= c1.min() + beam - 1e-9, (name, t)
        p0, p1 = d0.get_best_path(), d1.get_best_path()
        assert p0.ilabel.tolist() == p1.ilabel.tolist(), name
        assert p0.olabel.tolist() == p1.olabel.tolist(), name
        assert p0.graph_cost.tolist() == p1.graph_cost.tolist(), name
        assert p0.ac_cost.tolist() == p1.ac_cost.tolist(), name


def test_faster_equals_simple_on_ctc_topo(oracle):
    for V in (8, 32, 100):
        G = oracle.Graph(synth.ctc_topo(V))
        for utt in range(3):
            lp = synth.make_logprobs(60, V, utt, peak=5.0)
            f = oracle.FasterDecoder(G, beam=12.0, min_active=0)
            s = oracle.SimpleDecoder(G, beam=12.0)
            f.decode(lp)
            s.decode(lp)
            pf, ps = f.get_best_path(), s.get_best_path()
            assert pf.ok and ps.ok
            assert pf.ilabel.tolist() == ps.ilabel.tolist()
            assert pf.olabel.tolist() == ps.olabel.tolist()
            assert np.allclose(pf.ac_cost, ps.ac_cost, atol=1e-4)
            assert f.reached_final() and s.reached_final()
            assert s.final_relative_cost() == 0.0


@pytest.mark.parametrize("seed", range(6))
def test_lattice_shortest_path_equals_faster_best_path(oracle, seed):
    g = synth.random_graph(80 + 120 * seed, 3.0, 16, 0.2, 100 + seed)
    G = oracle.Graph(g)
    lp = synth.make_logprobs(100, 16, seed, peak=5.0)
    beam, lbeam = 6.0 + seed, 3.0 + seed
    f = oracle.FasterDecoder(G, beam=beam, min_active=0)
    f.decode(lp)
    pf = f.get_best_path()
    lat = oracle.LatticeSimpleDecoder(G, beam=beam, lattice_beam=lbeam, prune_interval=25)
    assert lat.decode(lp)
    pl = lat.get_best_path()
    assert pf.ok and pl.ok
    assert pf.osyms.tolist() == pl.osyms.tolist()
    assert pf.isyms.tolist() == pl.isyms.tolist()
    assert abs(sum(pf.total_cost) - sum(pl.total_cost)) < 1e-3
    raw = lat.get_raw_lattice()
    assert raw.ok and raw.num_states > 0 and len(raw.final_state) > 0
    assert lat.num_frames_decoded() == 100
    # lattice is frame-major and every arc goes forward (or stays, for epsilons)
    assert (raw.state_frame[raw.dst] >= raw.state_frame[raw.src]).all()
    em = raw.ilabel != 0
    assert (raw.state_frame[raw.dst][em] == raw.state_frame[raw.src][em] + 1).all()


def test_streaming_chunks_equal_one_shot(oracle):
    g = synth.hl_graph(V=20, n_words=60, seed=3)
    G = oracle.Graph(g)
    lp = synth.make_logprobs(50, 20, 1, peak=6.0, lexicon=g["lexicon"])
    for policy in (0, 1):
        a = oracle.FasterDecoder(G, beam=8.0, policy=policy)
        a.decode(lp)
        b = oracle.FasterDecoder(G, beam=8.0, policy=policy)
        b.init_decoding()
        # DecodableCtc(offset): a chunk object continues the frame index
        # (/root/reference/kaldi-decoder/csrc/decodable-ctc.cc:28-31)
        for start in range(0, 50, 7):
            b.advance_decoding(lp[start:start + 7], offset=start)
        assert b.num_frames_decoded() == 50
        c = oracle.FasterDecoder(G, beam=8.0, policy=policy)
        c.init_decoding()
        while c.num_frames_decoded() < 50:
            c.advance_decoding(lp, max_num_frames=3)
        pa, pb, pc = a.get_best_path(), b.get_best_path(), c.get_best_path()
        assert pa.ilabel.tolist() == pb.ilabel.tolist() == pc.ilabel.tolist()
        assert pa.ac_cost.tolist() == pb.ac_cost.tolist() == pc.ac_cost.tolist()


def test_error_behaviour(oracle):
    G = oracle.Graph(synth.ctc_topo(4))
    d = oracle.FasterDecoder(G)
    lp = synth.make_logprobs(3, 4, 0)
    with pytest.raises(RuntimeError, match="InitDecoding"):
        d.advance_decoding(lp)  # faster-decoder.cc:128-129
    for kw in ({"hash_ratio": 0.5}, {"max_active": 1}, {"min_active": -1},
               {"min_active": 30, "max_active": 30}):
        with pytest.raises(RuntimeError):  # faster-decoder.cc:24-28
            oracle.FasterDecoder(G, **kw)
    with pytest.raises(RuntimeError):  # lattice-simple-decoder.h:63-66
        oracle.LatticeSimpleDecoder(G, beam=0.0)
    lat = oracle.LatticeSimpleDecoder(G)
    lat.decode(lp)
    with pytest.raises(RuntimeError, match="FinalizeDecoding"):
        lat.get_raw_lattice(use_final_probs=False)  # lattice-simple-decoder.cc:594-597


def test_no_final_state_and_use_final_probs(oracle):
    # one non-final sink: ReachedFinal() false, best path still returned
    arcs = [(0, 1, 1, 5, 0.5), (1, 1, 1, 0, 0.1), (1, 2, 2, 6, 0.2), (2, 2, 2, 0, 0.0)]
    g = synth.graph_from_arcs(3, 0, arcs, {})
    G = oracle.Graph(g)
    lp = np.log(np.array([[0.9, 0.1], [0.5, 0.5], [0.1, 0.9]], np.float32))
    d = oracle.FasterDecoder(G, beam=20.0)
    d.decode(lp)
    assert not d.reached_final()
    p = d.get_best_path()
    assert p.ok and p.final_weight == (0.0, 0.0)
    assert p.osyms.tolist() == [5, 6]
    # now make state 2 final with cost 0.7
    g2 = synth.graph_from_arcs(3, 0, arcs, {2: 0.7})
    d = oracle.FasterDecoder(oracle.Graph(g2), beam=20.0)
    d.decode(lp)
    assert d.reached_final()
    assert d.get_best_path(True).final_weight[0] == np.float32(0.7)
    assert d.get_best_path(False).final_weight == (0.0, 0.0)


def test_max_active_and_min_active_paths(oracle):
    # exercise the nth_element branches (faster-decoder.cc:297-335); 1-best must
    # still match the unconstrained search when the constraint is loose enough
    g = synth.hl_graph(V=30, n_words=200, seed=5)
    G = oracle.Graph(g)
    lp = synth.make_logprobs(60, 30, 2, peak=8.0, lexicon=g["lexicon"])
    base = oracle.FasterDecoder(G, beam=10.0, min_active=0)
    base.decode(lp)
    ref = base.get_best_path()
    for policy in (0, 1):
        for kw in ({"max_active": 50}, {"min_active": 20}, {"max_active": 200, "min_active": 20}):
            d = oracle.FasterDecoder(G, beam=10.0, policy=policy, **kw)
            d.decode(lp)
            p = d.get_best_path()
            assert p.ok
            if "max_active" not in kw or kw["max_active"] >= 200:
                assert p.osyms.tolist() == ref.osyms.tolist(), (policy, kw)
        st = d.stats()
        assert st["frames"] == 60 and st["arcs_scanned"] > 0
    tight = oracle.FasterDecoder(G, beam=10.0, max_active=5, min_active=0)
    tight.decode(lp)
    assert tight.stats()["tokens_expanded"] <= 60 * 6
