"""Hand-derived cases for the three behaviours DESIGN section 2 listed as UNPINNED after round 3
-- places where the reference's result depends on a tolerance or on the order of its token
list, and where the order-free decoder (the canonical oracle, the GPU) deliberately computes
the exact fixpoint instead.  Each case is worked out in the comments from the cited lines of
/root/reference/kaldi-decoder/csrc/lattice-simple-decoder.cc, every number a dyadic rational;
the oracle's reference-order restatement (policy 0) must reproduce the reference's side of it,
the canonical oracle and the GPU the other, and the tests state the difference between the
two exactly (the canonical lattice is the reference's minus the links named here).

  I   PruneForwardLinksFinal's loop ends on ApproxEqual(old, new, 1e-5) (:512): a change of
      one part in 10^5 does not count as a change, so a token listed BEFORE the one that moved
      keeps a link that it priced one pass too early.
  J   ProcessNonemitting's queue is LIFO (:155-189): a diamond of epsilon arcs makes the
      reference expand a token twice (delete its links, generate them again); results are the
      same in any order, the count of links created is not.
  K   PruneForwardLinks inside ONE frame stops when no extra cost moved by more than delta
      (:286-290): with a large delta (prune_scale 0.75) a token listed before the token it
      links to keeps the stale price, and the un-finalized lattice holds two links whose exact
      cost is beyond lattice_beam.  FinalizeDecoding (delta 0) removes them.
"""
import numpy as np
import pytest

from kaldi_decoder import synth

INF = float("inf")

# ---------------------------------------------------------------------------------------
# Graph I.  V = 3, one frame, log-likelihoods 0.  beam 16384, lattice_beam 8193.0625.
#   state 0 (start)   a0 1:0/0.125 -> 1 (B)   a1 2:0/1 -> 2 (A)   a2 3:0/0 -> 3 (C)
#   state 3 (C, final 0)   a3 0:0/2 -> 2 (A)                      (epsilon)
#   state 2 (A)            a4 0:0/8191.125 -> 1 (B)               (epsilon)
#   state 1 (B, final 0)
# ProcessEmitting (:364-402) walks state 0's arcs in order: tokens B (0.125), A (1), C (0) are
# created in that order and PREPENDED to the frame's list (:95-96): the list reads C, A, B.
# ProcessNonemitting (:122-191), cutoff 0 + 16384: C -a3-> A at 0 + 2 = 2, not below A's 1: no
# change, but the link is made (:175-181); A -a4-> B at 1 + 8191.125 = 8192.125, not below
# 0.125: link made.  No token changes, whatever the queue's order.
# FinalizeDecoding -> PruneForwardLinksFinal (:425-520): final costs {C: 0, B: 0},
# final_best_cost = min(0 + 0, 0.125 + 0) = 0; every extra_cost starts at 0 (:91-95).
#   link_extra = next.extra + ((tok.tot + ac + graph) - next.tot) (:472-475):
#     C->A: A.extra + ((0 + 0 + 2) - 1)        = A.extra + 1
#     A->B: B.extra + ((1 + 0 + 8191.125) - 0.125) = B.extra + 8192
#   pass 1 (list order C, A, B):
#     C: own 0 + 0 - 0 = 0; C->A = 0 + 1 = 1 <= beam kept; extra min(0, 1) = 0   (was 0: same)
#     A: own 1 + inf = inf; A->B = 0 + 8192 kept; extra 8192                   (was 0: changed)
#     B: own 0.125; extra 0.125                                                (was 0: changed)
#   pass 2:
#     C: C->A = 8192 + 1 = 8193 <= 8193.0625 kept; extra 0                      (same)
#     A: A->B = 0.125 + 8192 = 8192.125 <= beam kept; extra 8192.125;  ApproxEqual(8192,
#        8192.125, 1e-5): |diff| 0.125 <= 1e-5 * 16384.125 = 0.1638...  -> NOT a change (:512)
#     B: 0.125                                                                  (same)
#   nothing "changed": the loop ends.  C->A stays, priced 8193 -- although A's extra cost is
#   8192.125 now and the link's exact cost 8193.125 is beyond lattice_beam: a third pass would
#   excise it (:476).  Frame 0 (PruneForwardLinks, delta 0): 0->B 0.125, 0->A 8192.125, 0->C 0,
#   all kept.  The reference's lattice: FIVE arcs.  The exact fixpoint: four (no C->A).


def graph_i():
    arcs = [(0, 1, 1, 0, 0.125), (0, 2, 2, 0, 1.0), (0, 3, 3, 0, 0.0), (3, 2, 0, 0, 2.0),
            (2, 1, 0, 0, 8191.125)]
    return synth.graph_from_arcs(4, 0, arcs, {3: 0.0, 1: 0.0})


LP_I = np.zeros((1, 3), np.float32)
OPTS_I = dict(beam=16384.0, lattice_beam=8193.0625)
ARCS_I = {
    "0B": (0, 0, 1, 1, 1, 0, 0.125, 0.0), "0A": (0, 0, 1, 2, 2, 0, 1.0, 0.0), "0C": (0, 0, 1, 3, 3, 0, 0.0, 0.0),
    "CA": (1, 3, 1, 2, 0, 0, 2.0, 0.0), "AB": (1, 2, 1, 1, 0, 0, 8191.125, 0.0),
}


def test_oracle_approx_equal_ends_the_final_pruning_one_pass_early(oracle):
    G = oracle.Graph(graph_i())
    d = oracle.LatticeSimpleDecoder(G, **OPTS_I)
    assert d.decode(LP_I)
    lat = d.get_raw_lattice()
    assert lat.arc_set() == set(ARCS_I.values())  # C->A survives at its stale price
    assert d.final_relative_cost() == 0.0
    # the order-free decoder iterates to the exact fixpoint: C->A (8193.125 > 8193.0625) is gone
    c = oracle.CanonLatticeDecoder(G, **OPTS_I)
    c.decode(LP_I)
    assert c.get_raw_lattice().arc_set() == {v for k, v in ARCS_I.items() if k != "CA"}
    # one sixteenth more beam and the link is inside either way; one eighth less and A->B
    # itself (8192.125) is out, A loses its last link, and both links into and out of A go
    for lbeam, gone in ((8193.125, ()), (8192.0625, ("CA", "AB", "0A"))):
        want = {v for k, v in ARCS_I.items() if k not in gone}
        for dec in (oracle.LatticeSimpleDecoder(G, beam=16384.0, lattice_beam=lbeam),
                    oracle.CanonLatticeDecoder(G, beam=16384.0, lattice_beam=lbeam)):
            dec.decode(LP_I)
            assert dec.get_raw_lattice().arc_set() == want, lbeam


# ---------------------------------------------------------------------------------------
# Graph J: a diamond of epsilon arcs below the start state.  V = 1, one frame, log-likelihood 0.
#   state 0 (start)  a0 0:0/1 -> 1 (X)      a1 0:0/0.25 -> 2 (Y)
#   state 1 (X)      a2 0:0/0 -> 3 (Z)
#   state 2 (Y)      a3 0:0/1.5 -> 3 (Z)
#   state 3 (Z)      a4 0:0/0.5 -> 4 (W)
#   state 4 (W)      a5 1:0/0 -> 5 (final 0)
# InitDecoding -> ProcessNonemitting (:122-191), beam 16, queue = [0]; the queue is a vector
# popped from the BACK (:155-156):
#   pop 0:  a0 -> X new at 1 (link 1, push X);  a1 -> Y new at 0.25 (link 2, push Y)   queue [X, Y]
#   pop Y:  a3 -> Z new at 1.75 (link 3, push Z)                                        queue [X, Z]
#   pop Z:  a4 -> W new at 2.25 (link 4; W has no epsilon arcs: not pushed, :183)       queue [X]
#   pop X:  a2 -> Z at 1 < 1.75: changed (link 5, push Z)                               queue [Z]
#   pop Z:  its links are deleted (:164-165) and made again: a4 -> W at 1.5 < 2.25 (link 6)
# SIX epsilon links created, five standing (a first-in-first-out queue would expand X and Y
# before Z, and Z once: five created).  Tokens {0: 0, X: 1, Y: 0.25, Z: 1, W: 1.5}.
# Frame 0: W -a5-> 5 at 1.5 (link 7).  Frame 1 {5: 1.5}, final.
# FinalizeDecoding, lattice_beam 0.5: Y->Z: 0 + ((0.25 + 1.5) - 1) = 0.75 > 0.5 excised, Y has
# no link left (extra +inf, :294-295), 0->Y goes with it: four arcs.  lattice_beam 1: all six.


def graph_j():
    arcs = [(0, 1, 0, 0, 1.0), (0, 2, 0, 0, 0.25), (1, 3, 0, 0, 0.0), (2, 3, 0, 0, 1.5),
            (3, 4, 0, 0, 0.5), (4, 5, 1, 0, 0.0)]
    return synth.graph_from_arcs(6, 0, arcs, {5: 0.0})


LP_J = np.zeros((1, 1), np.float32)
ARCS_J = {
    "0X": (0, 0, 0, 1, 0, 0, 1.0, 0.0), "0Y": (0, 0, 0, 2, 0, 0, 0.25, 0.0), "XZ": (0, 1, 0, 3, 0, 0, 0.0, 0.0),
    "YZ": (0, 2, 0, 3, 0, 0, 1.5, 0.0), "ZW": (0, 3, 0, 4, 0, 0, 0.5, 0.0), "W5": (0, 4, 1, 5, 1, 0, 0.0, 0.0),
}


def test_oracle_lifo_queue_expands_the_bottom_of_a_diamond_twice(oracle):
    G = oracle.Graph(graph_j())
    for lbeam, gone in ((0.5, ("0Y", "YZ")), (1.0, ())):
        want = {v for k, v in ARCS_J.items() if k not in gone}
        d = oracle.LatticeSimpleDecoder(G, beam=16.0, lattice_beam=lbeam)
        assert d.decode(LP_J)
        assert d.links_created() == 7  # six epsilon links (one of them made twice) + W->5
        assert d.get_raw_lattice().arc_set() == want
        c = oracle.CanonLatticeDecoder(G, beam=16.0, lattice_beam=lbeam)
        c.decode(LP_J)
        assert c.links_created() == 6  # every link once, from its source's final cost
        assert c.get_raw_lattice().arc_set() == want
        # the best path: 0 -> X -> Z -> W -> 5, graph costs 1 + 0 + 0.5 + 0
        assert c.get_best_path().graph_cost.tolist() == d.get_best_path().graph_cost.tolist() == [1.0, 0.0, 0.5, 0.0]


# ---------------------------------------------------------------------------------------
# Graph K.  V = 1, two frames, log-likelihoods 0.  beam 16, lattice_beam 4, prune_interval 1,
# prune_scale 0.75 (delta = 3, :58) or 0.1 (delta = 0.4).
#   state 0 (S, start)  a0 0:0/1 -> 1 (B)   a1 0:0/1 -> 2 (A)   a2 1:0/0 -> 3 (N)
#   state 2 (A)         a3 0:0/2.5 -> 1 (B)
#   state 1 (B)         a4 1:0/1 -> 3 (N)
#   state 3 (N)         a5 1:0/0 -> 4 (F, final 0)
# InitDecoding: pop S: a0 -> B new at 1 (B has no epsilon arcs: not queued), a1 -> A new at 1
# (queued); pop A: a3 -> B at 3.5, not below 1: link made, nothing changes.  Tokens created S,
# B, A; the list reads A, B, S.
# Frame 0: S -a2-> N at 0; B -a4-> N at 1 + 1 = 2 (link made, :393-398; N keeps 0).
# Top of the next iteration, NumFramesDecoded() = 1: PruneActiveTokens(delta) (:57-59), f = 0
# (never pruned), the tokens of frame 1 carry extra_cost 0:
#   pass 1, list order A, B, S (:247-292):
#     A: A->B = B.extra (still 0) + ((1 + 0 + 2.5) - 1) = 2.5 <= 4 kept; extra 2.5
#     B: B->N = 0 + ((1 + 0 + 1) - 0) = 2 kept; extra 2
#     S: S->B = 2 + 0 = 2; S->A = 2.5 + 0 = 2.5 kept; S->N = 0; extra 0
#   delta 3: no extra cost moved by more than 3 (2.5, 2, 0 against 0): `changed` stays false
#     (:286-288), the loop ends -- A->B keeps the price it got from B's STALE extra cost; its
#     exact cost is 2 + 2.5 = 4.5 > lattice_beam.
#   delta 0.4: pass 2: A->B = 2 + 2.5 = 4.5 > 4 excised (:264); A has no link left: +inf;
#     S->A = +inf: excised.
# Frame 1: N -a5-> F.  GetRawLattice before FinalizeDecoding: delta 3: all SIX links; delta 0.4:
# four (A stays listed -- PruneTokensForFrame is not run on frame 0 before FinalizeDecoding,
# :216-219 -- without arcs).  FinalizeDecoding (delta 0, :414): four arcs either way.


def graph_k():
    arcs = [(0, 1, 0, 0, 1.0), (0, 2, 0, 0, 1.0), (0, 3, 1, 0, 0.0), (2, 1, 0, 0, 2.5),
            (1, 3, 1, 0, 1.0), (3, 4, 1, 0, 0.0)]
    return synth.graph_from_arcs(5, 0, arcs, {4: 0.0})


LP_K = np.zeros((2, 1), np.float32)
ARCS_K = {
    "SB": (0, 0, 0, 1, 0, 0, 1.0, 0.0), "SA": (0, 0, 0, 2, 0, 0, 1.0, 0.0), "SN": (0, 0, 1, 3, 1, 0, 0.0, 0.0),
    "AB": (0, 2, 0, 1, 0, 0, 2.5, 0.0), "BN": (0, 1, 1, 3, 1, 0, 1.0, 0.0), "NF": (1, 3, 2, 4, 1, 0, 0.0, 0.0),
}
EXACT_K = {v for k, v in ARCS_K.items() if k not in ("SA", "AB")}


def test_oracle_a_large_delta_ends_a_frames_pruning_before_its_fixpoint(oracle):
    G = oracle.Graph(graph_k())
    d = oracle.LatticeSimpleDecoder(G, beam=16.0, lattice_beam=4.0, prune_interval=1, prune_scale=0.75)
    d.decode_nofinal(LP_K)
    lat = d.get_raw_lattice(True)
    assert lat.arc_set() == set(ARCS_K.values())  # A->B and S->A at their stale prices
    assert lat.state_set() == {(0, 0), (0, 1), (0, 2), (1, 3), (2, 4)}
    d.finalize_decoding()
    assert d.get_raw_lattice(True).arc_set() == EXACT_K
    d = oracle.LatticeSimpleDecoder(G, beam=16.0, lattice_beam=4.0, prune_interval=1, prune_scale=0.1)
    d.decode_nofinal(LP_K)
    lat = d.get_raw_lattice(True)
    assert lat.arc_set() == EXACT_K
    assert (0, 2) in lat.state_set()  # A is still listed (frame 0 is not token-pruned yet)
    c = oracle.CanonLatticeDecoder(G, beam=16.0, lattice_beam=4.0)
    c.decode(LP_K)
    assert c.get_raw_lattice().arc_set() == EXACT_K


# ---- the GPU: the order-free side of each case, through the C ABI ------------------------


@pytest.mark.gpu
def test_gpu_final_pruning_runs_to_the_exact_fixpoint():
    from kaldi_decoder import cabi
    G = cabi.Graph(graph_i())
    for lbeam, gone in ((8193.0625, ("CA",)), (8193.125, ()), (8192.0625, ("CA", "AB", "0A"))):
        lb = cabi.LatticeBatch(G, max_batch=1, beam=16384.0, lattice_beam=lbeam, min_active=0)
        lb.decode_host([LP_I])
        lat = lb.raw_lattices()[0]
        assert lat.arc_set() == {v for k, v in ARCS_I.items() if k not in gone}, lbeam
        assert lb.final_relative_cost().tolist() == [0.0]


@pytest.mark.gpu
def test_gpu_diamond_of_epsilon_arcs():
    from kaldi_decoder import cabi
    G = cabi.Graph(graph_j())
    for lbeam, gone in ((0.5, ("0Y", "YZ")), (1.0, ())):
        lb = cabi.LatticeBatch(G, max_batch=1, beam=16.0, lattice_beam=lbeam, min_active=0)
        lb.decode_host([LP_J])
        assert lb.raw_lattices()[0].arc_set() == {v for k, v in ARCS_J.items() if k not in gone}
        st = dict(zip(cabi.STAT_NAMES, lb.stats()[0].tolist()))
        assert st["links"] == 6  # every link once (the reference makes Z's link twice)
    bt = cabi.Batch(G, max_batch=1, beam=16.0, min_active=0)
    bt.decode_host([LP_J])
    assert bt.best_paths()[0].state.tolist() == [1, 3, 4, 5]


@pytest.mark.gpu
@pytest.mark.parametrize("prune_scale", [0.75, 0.1])
def test_gpu_periodic_pruning_runs_every_frame_to_its_fixpoint(prune_scale):
    from kaldi_decoder import cabi
    lb = cabi.LatticeBatch(cabi.Graph(graph_k()), max_batch=1, beam=16.0, lattice_beam=4.0, min_active=0,
                           prune_interval=1, prune_scale=prune_scale)
    lb.init_decoding()
    lb.advance_host([LP_K])
    lat = lb.raw_lattices()[0]
    # (the reference at prune_scale 0.75 still holds S->A and A->B here: documented difference)
    assert lat.arc_set() == EXACT_K
    assert lat.state_set() == {(0, 0), (0, 1), (0, 2), (1, 3), (2, 4)}
    lb.finalize_decoding()
    lat = lb.raw_lattices()[0]
    assert lat.arc_set() == EXACT_K and (0, 2) not in lat.state_set()
