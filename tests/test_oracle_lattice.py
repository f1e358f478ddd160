"""Lattice decoders in the oracle: the reference-order LatticeSimpleDecoder
restatement (policy 0) against the canonical, order-free lattice decoder
(policy 1) that the GPU implements.  Lattices are compared numbering-free, as
sets of arcs keyed by (frame, graph state) of both ends."""
import numpy as np
import pytest

import golden_util as gu
from kaldi_decoder import synth


def _cases():
    for seed in range(4):
        yield (synth.random_graph(100 + 150 * seed, 3.0, 16, 0.2, 200 + seed), None, 16,
               6.0 + seed, 3.0 + seed)
    g = synth.hl_graph(V=30, n_words=200, seed=5)
    yield g, g["lexicon"], 30, 10.0, 6.0
    g = synth.hlg_graph(V=30, n_words=100, n_hist=100, avg_succ=6, seed=6)
    yield g, g["lexicon"], 30, 9.0, 5.0
    yield synth.ctc_topo(16), None, 16, 8.0, 4.0


@pytest.mark.parametrize("prune_interval", [1, 25])
def test_canonical_lattice_is_the_reference_lattice_minus_extras(oracle, prune_interval):
    for g, lex, V, beam, lbeam in _cases():
        G = oracle.Graph(g)
        for utt in range(2):
            lp = synth.make_logprobs(60, V, utt, peak=5.0, lexicon=lex)
            ref = oracle.LatticeSimpleDecoder(G, beam=beam, lattice_beam=lbeam,
                                              prune_interval=prune_interval)
            can = oracle.CanonLatticeDecoder(G, beam=beam, lattice_beam=lbeam)
            assert ref.decode(lp) == can.decode(lp)
            assert can.num_frames_decoded() == ref.num_frames_decoded() == 60
            rl, cl = ref.get_raw_lattice(), can.get_raw_lattice()
            assert rl.ok and cl.ok
            ra, ca = rl.arc_set(), cl.arc_set()
            assert ca <= ra  # every canonical arc is a reference arc, costs bit-equal
            assert cl.state_set() <= rl.state_set()
            assert cl.final_set() <= rl.final_set()
            assert ref.final_relative_cost() == can.final_relative_cost()
            # 1-best through either lattice is the same path
            pr, pc = ref.get_best_path(), can.get_best_path()
            assert pr.ok and pc.ok
            assert pr.isyms.tolist() == pc.isyms.tolist() and pr.osyms.tolist() == pc.osyms.tolist()
            assert abs(sum(pr.total_cost) - sum(pc.total_cost)) < 1e-3
            # reference-only arcs are the order-dependent extras: few, and never on the best path
            assert len(ra - ca) <= 0.15 * len(ra) + 5
            # exactly: a reference arc lies below (frame best + beam), the value the running
            # cutoff ends the frame at, iff the order-free decoder created that link as well
            frame_best = [can.frame_tokens(f)[1].min() for f in range(61)]
            gu.assert_extras_beyond_cutoff(gu.lattice_rows(rl), can.created_links(), frame_best,
                                           beam, g["start"], (prune_interval, utt))


def test_canonical_lattice_contains_faster_decoder_best_path(oracle):
    g = synth.random_graph(300, 3.0, 16, 0.2, 77)
    G = oracle.Graph(g)
    lp = synth.make_logprobs(80, 16, 3, peak=5.0)
    f = oracle.FasterDecoder(G, beam=9.0, min_active=0, policy=1)
    f.decode(lp)
    pf = f.get_best_path()
    can = oracle.CanonLatticeDecoder(G, beam=9.0, lattice_beam=5.0)
    can.decode(lp)
    pc = can.get_best_path()
    assert pf.osyms.tolist() == pc.osyms.tolist() and pf.isyms.tolist() == pc.isyms.tolist()
    assert abs(sum(pf.total_cost) - sum(pc.total_cost)) < 1e-3


def test_lattice_faster_options(oracle):
    # max_active / min_active: the LatticeFasterDecoder configuration
    g = synth.hl_graph(V=30, n_words=200, seed=5)
    G = oracle.Graph(g)
    lp = synth.make_logprobs(50, 30, 1, peak=7.0, lexicon=g["lexicon"])
    base = oracle.CanonLatticeDecoder(G, beam=10.0, lattice_beam=6.0)
    base.decode(lp)
    loose = oracle.CanonLatticeDecoder(G, beam=10.0, lattice_beam=6.0, max_active=100000, min_active=5)
    loose.decode(lp)
    assert base.get_raw_lattice().arc_set() == loose.get_raw_lattice().arc_set()
    tight = oracle.CanonLatticeDecoder(G, beam=10.0, lattice_beam=6.0, max_active=20)
    tight.decode(lp)
    assert tight.links_created() < base.links_created()
    assert tight.get_best_path().ok
    with pytest.raises(RuntimeError):
        oracle.CanonLatticeDecoder(G, beam=10.0, lattice_beam=0.0)
