"""Known-answer tests, second set: cases of several frames, worked out BY HAND from the
reference source (not from the oracle), for the places tests/test_kat.py's two-frame cases
do not reach -- the order in which a HashList is iterated, `nth_element` on tied costs with
the double -> float narrowing of `tmp_array_`, PruneActiveTokens stopping early, an
utterance in which no token survives.  Weights and log-likelihoods are dyadic rationals
(float and double arithmetic exact).  "ref:" = /root/reference/kaldi-decoder/csrc/.
"""
import numpy as np
import pytest

from kaldi_decoder import synth

INF = float("inf")
INT_MAX = 2**31 - 1


def _check_path(p, want, fw, tag):
    assert p.ok, tag
    assert p.ilabel.tolist() == want["ilabel"], tag
    assert p.olabel.tolist() == want["olabel"], tag
    assert p.state.tolist() == want["state"], tag
    assert p.graph_cost.tolist() == want["graph"], tag
    assert p.ac_cost.tolist() == want["ac"], tag
    assert tuple(float(x) for x in p.final_weight) == fw, tag


# ---------------------------------------------------------------------------------------
# Graph D: a token that exists only because of the ORDER in which the reference walks its
# HashList -- created under a running cutoff that a later token tightens -- sits in the
# next frame's token list (policy 0) and is never expanded; the order-free decoder
# (policy 1, the GPU) does not have it; the 1-best is the same.
#   FasterDecoder, beam 2, max_active INT_MAX, min_active 0 (GetCutoff's fast path,
#   faster-decoder.cc:249-272: weight_cutoff = best + 2, adaptive_beam = 2).  V = 3.
#   state 0 (start)  a0 1:1/1.0  -> 1    a1 2:2/1.5 -> 2    a2 3:3/1.25 -> 3
#   state 1          a3 1:4/2.0  -> 4
#   state 2          a4 1:5/3.0  -> 5
#   state 3          a5 1:6/0.25 -> 6
#   state 4          a6 1:0/0.5  -> 7
#   state 5          a7 1:0/0    -> 7
#   state 6          a8 1:7/0.25 -> 7    a9 2:8/0.5 -> 8
#   state 7 (final 0)      a10 1:0/1.0 -> 7
#   state 8 (final 0.125)  a11 1:0/0   -> 8
# log-likelihoods: frames 0-2 all zero; frame 3: [-0.25, 0, 0].
#
# The HashList has 1000 buckets from the constructor (faster-decoder.cc:31) and bucket =
# state % 1000 (hash-list-inl.h:129): every state has a bucket of its own, so the list
# order is the order in which states were first inserted (:150-163 appends a newly occupied
# bucket's element at the tail).
#  InitDecoding: {0: 0}.
#  frame 0: weight_cutoff 2.  Best token's arcs first (:176-189): new weights 1.0, 1.5, 1.25
#    -> next_weight_cutoff = 1.0 + 2 = 3.0.  All three below it (:211): inserted in arc
#    order.  List: [1: 1.0, 2: 1.5, 3: 1.25].
#  frame 1: best 1.0 (state 1, the first minimum, :251-257), weight_cutoff 3.0.  Best token's
#    arc a3: 2.0 + 1.0 = 3.0 -> next_weight_cutoff = 5.0.  Then the list, in order:
#      state 1: a3 -> 4: 3.0 < 5.0, inserted; 3.0 + 2 = 5.0 is not < 5.0 (:215).
#      state 2: a4 -> 5: 3.0 + 1.5 = 4.5 < 5.0: INSERTED (the running cutoff is still 5.0).
#      state 3: a5 -> 6: 0.25 + 1.25 = 1.5 < 5.0, inserted; cutoff := 1.5 + 2 = 3.5.
#    Returned cutoff 3.5.  Reference-order frontier: {4: 3.0, 5: 4.5, 6: 1.5}: token 5 lies
#    beyond the cutoff the frame ends at -- had state 3 been walked before state 2 it would
#    not exist.  Order-free frontier (cost < 3.5): {4: 3.0, 6: 1.5}.
#  frame 2: list [4, 5, 6]: best 1.5 (state 6), weight_cutoff 3.5.  Best token's arcs:
#    a8: 1.75 -> cutoff 3.75; a9: 2.0 -> 4.0, no.  Then:
#      state 4 (3.0 < 3.5): a6 -> 7: 3.5 < 3.75, inserted.
#      state 5 (4.5 is not < 3.5, :202): not expanded -- the extra token ends here.
#      state 6: a8 -> 7: 1.75 < 3.75: state 7 has 3.5, strictly worse: replaced (:220-224);
#               a9 -> 8: 2.0, inserted.
#    Frontier {7: 1.75, 8: 2.0}, both policies.
#  frame 3 (ll[ilabel 1] = -0.25): best 1.75, weight_cutoff 3.75.  a10: 1.0 + 1.75 + 0.25 =
#    3.0 -> cutoff 5.0;  state 7: a10 -> 7: 3.0;  state 8: a11 -> 8: 0 + 2.0 + 0.25 = 2.25,
#    cutoff 4.25.  Frontier {7: 3.0, 8: 2.25}.
#  ReachedFinal.  GetBestPath (:362-384): 7: 3.0 + 0 = 3.0;  8: 2.25 + 0.125 = 2.375 -> state 8:
#    a2, a5, a9, a11.  ac = float(cost - prev.cost) - graph (:394-396): 0, 0, 0, 0.25.
def graph_d():
    arcs = [(0, 1, 1, 1, 1.0), (0, 2, 2, 2, 1.5), (0, 3, 3, 3, 1.25), (1, 4, 1, 4, 2.0),
            (2, 5, 1, 5, 3.0), (3, 6, 1, 6, 0.25), (4, 7, 1, 0, 0.5), (5, 7, 1, 0, 0.0),
            (6, 7, 1, 7, 0.25), (6, 8, 2, 8, 0.5), (7, 7, 1, 0, 1.0), (8, 8, 1, 0, 0.0)]
    return synth.graph_from_arcs(9, 0, arcs, {7: 0.0, 8: 0.125})


LP_D = np.array([[0, 0, 0], [0, 0, 0], [0, 0, 0], [-0.25, 0, 0]], np.float32)
OPTS_D = dict(beam=2.0, min_active=0, max_active=INT_MAX)
FRONTIERS_D = {
    0: [{0: 0.0}, {1: 1.0, 2: 1.5, 3: 1.25}, {4: 3.0, 5: 4.5, 6: 1.5}, {7: 1.75, 8: 2.0},
        {7: 3.0, 8: 2.25}],
    1: [{0: 0.0}, {1: 1.0, 2: 1.5, 3: 1.25}, {4: 3.0, 6: 1.5}, {7: 1.75, 8: 2.0},
        {7: 3.0, 8: 2.25}],
}
PATH_D = dict(ilabel=[3, 1, 2, 1], olabel=[3, 6, 8, 0], state=[3, 6, 8, 8],
              graph=[1.25, 0.25, 0.5, 0.0], ac=[0.0, 0.0, 0.0, 0.25])

# ---------------------------------------------------------------------------------------
# Graph E: max_active binds on TIED costs, and doubles that narrow onto the cutoff.
#   FasterDecoder, beam 8, max_active 2, min_active 0, beam_delta 0.5.  V = 5.  e = 2^-30.
#   state 0 (start)  a0 1:1/1.0 -> 1   a1 2:2/1.0 -> 2   a2 3:3/0.5 -> 3   a3 4:4/1.0 -> 4
#                    a4 5:5/2.0 -> 5   a5 2:6/1.0 -> 6
#   state 1  a6 1:0/0 -> 1        state 2  a7 1:12/0.25 -> 7     state 3  a8 1:13/0.5 -> 8
#   state 4  a9 1:0/0 -> 4        state 5  a10 1:0/0 -> 5        state 6  a11 1:16/0.125 -> 9
#   states 7, 8, 9 final 0.
# log-likelihoods: frame 0: [-e, +e, 0, 0, 0]; frame 1: zeros.
#  frame 0: one token: not more than max_active (:289); more than min_active = 0 ->
#    min_active_cutoff = best_cost = 0 (:312-313), not above beam_cutoff 8: adaptive_beam =
#    beam, weight_cutoff 8 (:331-334).  new_weight = arc.weight + tok.cost + ac_cost in DOUBLE
#    (:210): 1: 1 + e;  2: 1 - e;  3: 0.5;  4: 1.0;  5: 2.0;  6: 1 - e.  All below the running
#    cutoff (0.5 + 8).  List order [1, 2, 3, 4, 5, 6].
#  frame 1, GetCutoff: tmp_array_ is a vector<float> (faster-decoder.h:186): push_back
#    narrows (:278): [1, 1, 0.5, 1, 2, 1] (1 +- 2^-30 round to 1.0f).  6 > max_active:
#    nth_element at 2 (:291-295): sorted [0.5, 1, 1, 1, 1, 2] -> 1.0, whichever of the tied
#    elements lands there.  1.0 < beam_cutoff 8.5: adaptive_beam = 1.0 - 0.5 + 0.5 = 1.0,
#    weight_cutoff 1.0 (:297-303).
#    Expanded iff cost_ < 1.0 in double (:202): state 3 (0.5), states 2 and 6 (1 - e) -- THREE
#    tokens although max_active is 2; not 4 (1.0), not 1 (1 + e), all of which were "1.0" to
#    nth_element.
#    Best token (3): a8: 0.5 + 0.5 = 1.0 -> next_weight_cutoff 2.0.  Then in list order:
#      2: a7 -> 7: 0.25 + (1 - e) = 1.25 - e     3: a8 -> 8: 1.0     6: a11 -> 9: 1.125 - e
#    Frontier {7: 1.25 - e, 8: 1.0, 9: 1.125 - e}.
#  Best final token: 8 (1.0): a2, a8; graph 0.5, 0.5; ac 0, 0.
E = 2.0 ** -30


def graph_e():
    arcs = [(0, 1, 1, 1, 1.0), (0, 2, 2, 2, 1.0), (0, 3, 3, 3, 0.5), (0, 4, 4, 4, 1.0),
            (0, 5, 5, 5, 2.0), (0, 6, 2, 6, 1.0), (1, 1, 1, 0, 0.0), (2, 7, 1, 12, 0.25),
            (3, 8, 1, 13, 0.5), (4, 4, 1, 0, 0.0), (5, 5, 1, 0, 0.0), (6, 9, 1, 16, 0.125)]
    return synth.graph_from_arcs(10, 0, arcs, {7: 0.0, 8: 0.0, 9: 0.0})


LP_E = np.array([[-E, E, 0, 0, 0], [0, 0, 0, 0, 0]], np.float32)
OPTS_E = dict(beam=8.0, min_active=0, max_active=2, beam_delta=0.5)
FRONTIERS_E = [{0: 0.0}, {1: 1 + E, 2: 1 - E, 3: 0.5, 4: 1.0, 5: 2.0, 6: 1 - E},
               {7: 1.25 - E, 8: 1.0, 9: 1.125 - E}]
PATH_E = dict(ilabel=[3, 1], olabel=[3, 13], state=[3, 8], graph=[0.5, 0.5], ac=[0.0, 0.0])

# ---------------------------------------------------------------------------------------
# Graph G: no token survives.  state 0 (start) a0 1:1/1.0 -> 1; state 1 (final 0) has no
# arcs.  Three frames of zeros: frame 0 leaves {1: 1.0}; in frame 1 the token has nothing
# to propagate along: the list is empty from then on (GetCutoff on an empty list: best_cost
# +inf, :244-272, nothing to walk).  ReachedFinal false (no token, :347-354); GetBestPath
# finds no best token and returns false (:367-386).
def graph_g():
    return synth.graph_from_arcs(2, 0, [(0, 1, 1, 1, 1.0)], {1: 0.0})


LP_G = np.zeros((3, 1), np.float32)
FRONTIERS_G = [{0: 0.0}, {1: 1.0}, {}, {}]


@pytest.mark.parametrize("policy", [0, 1])
def test_oracle_hashlist_order_extra_lives_one_frame(oracle, policy):
    d = oracle.FasterDecoder(oracle.Graph(graph_d()), policy=policy, **OPTS_D)
    d.init_decoding()
    for f, want in enumerate(FRONTIERS_D[policy]):
        if f:
            d.advance_decoding(LP_D, max_num_frames=1)
        s, c = d.tokens()
        assert dict(zip(s.tolist(), c.tolist())) == want, (policy, f)
    assert d.reached_final()
    _check_path(d.get_best_path(True), PATH_D, (0.125, 0.0), policy)
    _check_path(d.get_best_path(False), PATH_D, (0.0, 0.0), policy)


@pytest.mark.parametrize("policy", [0, 1])
def test_oracle_max_active_on_tied_and_narrowed_costs(oracle, policy):
    d = oracle.FasterDecoder(oracle.Graph(graph_e()), policy=policy, **OPTS_E)
    d.init_decoding()
    for f, want in enumerate(FRONTIERS_E):
        if f:
            d.advance_decoding(LP_E, max_num_frames=1)
        s, c = d.tokens()
        assert dict(zip(s.tolist(), c.tolist())) == want, (policy, f)
    st = d.stats()
    assert st["tokens_expanded"] == 1 + 3  # frame 1: three tokens pass a max_active of 2
    _check_path(d.get_best_path(True), PATH_E, (0.0, 0.0), policy)


@pytest.mark.parametrize("policy", [0, 1])
def test_oracle_no_token_survives(oracle, policy):
    d = oracle.FasterDecoder(oracle.Graph(graph_g()), policy=policy, beam=16.0, min_active=0)
    d.init_decoding()
    for f, want in enumerate(FRONTIERS_G):
        if f:
            d.advance_decoding(LP_G, max_num_frames=1)
        s, c = d.tokens()
        assert dict(zip(s.tolist(), c.tolist())) == want, (policy, f)
    assert not d.reached_final()
    assert not d.get_best_path(True).ok and not d.get_best_path(False).ok


# ---------------------------------------------------------------------------------------
# Graph F: PruneActiveTokens stops early.  LatticeSimpleDecoder, beam 16, lattice_beam 4,
# prune_interval 2, prune_scale 0.9 (delta = 3.6: ref lattice-simple-decoder.cc:58).  V = 4.
#   state 0 (start)   a0 1:10/1.0 -> 1 (A)     a1 2:20/3.0 -> 2 (Y)
#   state 1 (A, final 0.5)  a2 1:0/1.0 -> 1    a3 3:30/1.0 -> 3 (X)
#   state 2 (Y)       a4 3:40/1.0 -> 3 (X)
#   state 3 (X)       a5 4:50/4.0 -> 1 (A)
# log-likelihoods, 0 where an ilabel is "on", -64 where it is "off" (an off arc costs 64 more
# than anything else: beyond `cutoff` once any on-arc has been scored, and every token's first
# arc is on in the frames where the token exists, so no walk order creates a token from one):
#   frame 0: ilabels 1, 2 on     frame 1: 1, 3 on     frame 2: 1, 4 on     frames 3, 4: 1 on
# Tokens (tot_cost = cur + ac + graph, :383-385), frame-plus-one f:
#   f0 {0: 0}   f1 {A: 1, Y: 3}   f2 {A: 2, X: 2}   f3 {A: 3}   f4 {A: 4}   f5 {A: 5}
#   (X at f2: from A 1 + 1 = 2, from Y 3 + 1 = 4: the token keeps 2, both arcs are links,
#   :393-398; A at f3: from A 3, from X 2 + 4 = 6: both links.)
# Decode() prunes when NumFramesDecoded() % 2 == 0 (:57-59): at 0 (nothing), 2 and 4.
#  PruneActiveTokens at 2 (cur_frame_plus_one 2, tokens of f2 carry extra_cost 0):
#   f = 1 (never pruned: must_prune_forward_links true): link_extra_cost = next.extra +
#     ((tok.tot + ac + graph) - next.tot) (:257-260):  A->A 0,  A->X 0,  Y->X 0 + (3 + 1 - 2)
#     = 2 <= lattice_beam: kept.  extras {A: 0, Y: 2}; |2 - 0| = 2 is not > delta 3.6:
#     extra_costs_changed false (:286-290).
#   f = 0 (never pruned): 0->A 0;  0->Y: 2 + (0 + 3 - 3) = 2: kept.  extra {0: 0}.
#  PruneActiveTokens at 4 (tokens of f4 carry 0):
#   f = 3: A->A 0: extra 0, unchanged.
#   f = 2 (never pruned): A->A 0;  X->A: 0 + (2 + 4 - 3) = 3 <= 4: kept.  extras {A: 0, X: 3};
#     |3 - 0| = 3 is not > 3.6: extra_costs_changed FALSE, so must_prune_forward_links of
#     frame 1 stays false (:212-213) and f = 1, f = 0 are NOT visited (:208): link Y->X keeps
#     its stale cost 2 although its exact cost is now X.extra + 2 = 5 > lattice_beam.
#  The loop ends at 5 frames.  GetRawLattice before FinalizeDecoding: all NINE links.
#  FinalizeDecoding (delta 0, every frame, :407-420): Y->X: 3 + 2 = 5 > 4 excised; Y is left
#   without links, extra +inf, deleted (:294-295, :320-329); 0->Y excised with it: SEVEN.
#  (A sweep that went on below frame 2 at NumFramesDecoded() == 4 would show seven arcs
#   before FinalizeDecoding already.)
def graph_f():
    arcs = [(0, 1, 1, 10, 1.0), (0, 2, 2, 20, 3.0), (1, 1, 1, 0, 1.0), (1, 3, 3, 30, 1.0),
            (2, 3, 3, 40, 1.0), (3, 1, 4, 50, 4.0)]
    return synth.graph_from_arcs(4, 0, arcs, {1: 0.5})


OFF = -64.0
LP_F = np.array([[0, 0, OFF, OFF], [0, OFF, 0, OFF], [0, OFF, OFF, 0], [0, OFF, OFF, OFF],
                 [0, OFF, OFF, OFF]], np.float32)
OPTS_F = dict(beam=16.0, lattice_beam=4.0, prune_interval=2, prune_scale=0.9)
# (src frame, src state, dst frame, dst state, ilabel, olabel, graph, ac)
ARCS_F = {
    "0A": (0, 0, 1, 1, 1, 10, 1.0, 0.0), "0Y": (0, 0, 1, 2, 2, 20, 3.0, 0.0),
    "AA1": (1, 1, 2, 1, 1, 0, 1.0, 0.0), "AX": (1, 1, 2, 3, 3, 30, 1.0, 0.0),
    "YX": (1, 2, 2, 3, 3, 40, 1.0, 0.0), "AA2": (2, 1, 3, 1, 1, 0, 1.0, 0.0),
    "XA": (2, 3, 3, 1, 4, 50, 4.0, 0.0), "AA3": (3, 1, 4, 1, 1, 0, 1.0, 0.0),
    "AA4": (4, 1, 5, 1, 1, 0, 1.0, 0.0),
}
BEFORE_F = set(ARCS_F.values())
AFTER_F = {v for k, v in ARCS_F.items() if k not in ("0Y", "YX")}


def test_oracle_prune_active_tokens_stops_early(oracle):
    G = oracle.Graph(graph_f())
    d = oracle.LatticeSimpleDecoder(G, **OPTS_F)
    d.decode_nofinal(LP_F)
    assert d.num_frames_decoded() == 5
    lat = d.get_raw_lattice(True)
    assert lat.arc_set() == BEFORE_F
    assert lat.final_set() == {(1, 0.5)}
    d.finalize_decoding()
    assert d.get_raw_lattice(True).arc_set() == AFTER_F
    # prune_scale small enough that the change of 3 at frame 2 counts (delta 0.4): the sweep
    # at 4 goes on to frames 1 and 0 and the two links are gone before FinalizeDecoding
    d = oracle.LatticeSimpleDecoder(G, **{**OPTS_F, "prune_scale": 0.1})
    d.decode_nofinal(LP_F)
    assert d.get_raw_lattice(True).arc_set() == AFTER_F
    # the order-free decoder prunes in FinalizeDecoding only
    c = oracle.CanonLatticeDecoder(G, beam=16.0, lattice_beam=4.0)
    c.decode(LP_F)
    assert c.get_raw_lattice().arc_set() == AFTER_F


# ---------------------------------------------------------------------------------------
# Graph H: PruneForwardLinksFinal's "iterate until nothing changes" loop (ref:
# lattice-simple-decoder.cc:425-520) -- a link that is excised only in the SECOND pass,
# because the first pass read its destination's extra cost before that was computed (the
# frame's token list is not in topological order, :437-440).  One frame, V = 3,
# log-likelihoods -0.5 / -0.25 / -0.125 (acoustic costs 0.5 / 0.25 / 0.125), beam 16.
#   state 0 (start)  a0 1:1/1.0 -> 1 (B)   a1 2:2/0.5 -> 2 (A)   a2 3:3/0.375 -> 3 (C)
#   state 2 (A)      a3 eps:4/0.25 -> 1 (B)
#   final: B 0, A 2.0, C 0
# ProcessEmitting (:364-402) walks a0, a1, a2: tokens B 1.5, A 0.75, C 0.5, each PREPENDED to
# the frame's list (:95-96): the list is C, A, B.  ProcessNonemitting (:122-191): only state 2
# has an epsilon arc; A -> B costs 0.75 + 0.25 = 1.0 < 1.5: B's tot_cost becomes 1.0 (:104-107)
# and the link A -> B (graph 0.25, ac 0) is kept.  Links of the start token: ->B (1.0, 0.5),
# ->A (0.5, 0.25), ->C (0.375, 0.125).
# ComputeFinalCosts (:522-560): B 1.0, A 2.75, C 0.5: final_best_cost 0.5, best_cost 0.5,
# FinalRelativeCost 0.  Every token's extra_cost is still the 0 it was created with.
#  lattice_beam 0.5:
#   pass 1  C: 0.5 + 0 - 0.5 = 0, unchanged.   A: starts at 0.75 + 2 - 0.5 = 2.25; link A->B:
#           B.extra (STALE 0) + ((0.75 + 0 + 0.25) - 1.0) = 0: kept, extra 0, unchanged.
#           B: 1.0 + 0 - 0.5 = 0.5, not > 0.5: extra 0.5, changed.
#   pass 2  A: link A->B: 0.5 + 0 = 0.5, not > 0.5: kept; extra min(2.25, 0.5) = 0.5, changed.
#   pass 3  nothing changes.
#   PruneForwardLinks(0) (:228-305): ->B: 0.5 + ((0 + 0.5 + 1.0) - 1.0) = 1.0 > 0.5 EXCISED (B
#   is reached 0.5 cheaper through A);  ->A: 0.5 + 0 = 0.5 kept;  ->C: 0 kept.  No token has
#   extra +inf: four states, THREE arcs, finals B 0 / A 2 / C 0.
#  lattice_beam 0.375:
#   pass 1  A: link A->B at the stale 0: kept, extra 0.   B: 0.5 > 0.375: extra +inf, changed.
#   pass 2  A: link A->B: +inf > 0.375: EXCISED NOW; extra 2.25 > 0.375: +inf, changed.
#   pass 3  nothing changes.
#   PruneForwardLinks(0): ->B and ->A excised (+inf), ->C kept; PruneTokensForFrame(1)
#   (:310-334) deletes A and B: two states, ONE arc, final C 0.
def graph_h():
    arcs = [(0, 1, 1, 1, 1.0), (0, 2, 2, 2, 0.5), (0, 3, 3, 3, 0.375), (2, 1, 0, 4, 0.25)]
    return synth.graph_from_arcs(4, 0, arcs, {1: 0.0, 2: 2.0, 3: 0.0})


LP_H = np.array([[-0.5, -0.25, -0.125]], np.float32)
ARCS_H = {
    "0B": (0, 0, 1, 1, 1, 1, 1.0, 0.5), "0A": (0, 0, 1, 2, 2, 2, 0.5, 0.25),
    "0C": (0, 0, 1, 3, 3, 3, 0.375, 0.125), "AB": (1, 2, 1, 1, 0, 4, 0.25, 0.0),
}
# lattice_beam -> (arcs, finals) after FinalizeDecoding
AFTER_H = {
    0.5: ({ARCS_H[k] for k in ("0A", "0C", "AB")}, {(1, 0.0), (2, 2.0), (3, 0.0)}),
    0.375: ({ARCS_H["0C"]}, {(3, 0.0)}),
}


@pytest.mark.parametrize("lattice_beam", [0.5, 0.375])
def test_oracle_final_pruning_needs_a_second_pass(oracle, lattice_beam):
    G = oracle.Graph(graph_h())
    arcs, finals = AFTER_H[lattice_beam]
    d = oracle.LatticeSimpleDecoder(G, beam=16.0, lattice_beam=lattice_beam)
    d.decode_nofinal(LP_H)
    lat = d.get_raw_lattice(True)
    assert lat.arc_set() == set(ARCS_H.values())  # nothing pruned before FinalizeDecoding
    assert lat.final_set() == {(1, 0.0), (2, 2.0), (3, 0.0)}
    assert d.final_relative_cost() == 0.0
    d.finalize_decoding()
    lat = d.get_raw_lattice(True)
    assert lat.arc_set() == arcs and lat.final_set() == finals
    assert lat.num_states == 1 + len(finals)
    c = oracle.CanonLatticeDecoder(G, beam=16.0, lattice_beam=lattice_beam)
    c.decode(LP_H)
    lat = c.get_raw_lattice()
    assert lat.arc_set() == arcs and lat.final_set() == finals


# ---- the same cases through the C ABI on the GPU ---------------------------------------
@pytest.mark.gpu
def test_gpu_hashlist_order_extra(oracle):
    from kaldi_decoder import cabi
    bt = cabi.Batch(cabi.Graph(graph_d()), max_batch=1, **OPTS_D)
    bt.init_decoding()
    for f, want in enumerate(FRONTIERS_D[1]):
        if f:
            bt.advance_host([LP_D], max_num_frames=1)
        s, c = bt.frame_tokens(0, f)
        assert dict(zip(s.tolist(), c.tolist())) == want, f
    assert bool(bt.reached_final()[0])
    _check_path(bt.best_paths(True)[0], PATH_D, (0.125, 0.0), "gpu")
    _check_path(bt.best_paths(False)[0], PATH_D, (0.0, 0.0), "gpu")


@pytest.mark.gpu
def test_gpu_max_active_on_tied_and_narrowed_costs():
    from kaldi_decoder import cabi
    bt = cabi.Batch(cabi.Graph(graph_e()), max_batch=1, **OPTS_E)
    bt.init_decoding()
    for f, want in enumerate(FRONTIERS_E):
        if f:
            bt.advance_host([LP_E], max_num_frames=1)
        s, c = bt.frame_tokens(0, f)
        assert dict(zip(s.tolist(), c.tolist())) == want, f
    st = dict(zip(cabi.STAT_NAMES, bt.stats()[0].tolist()))
    assert st["tokens_expanded"] == 4
    _check_path(bt.best_paths(True)[0], PATH_E, (0.0, 0.0), "gpu")


@pytest.mark.gpu
def test_gpu_no_token_survives():
    from kaldi_decoder import cabi
    bt = cabi.Batch(cabi.Graph(graph_g()), max_batch=1, beam=16.0, min_active=0)
    bt.init_decoding()
    for f, want in enumerate(FRONTIERS_G):
        if f:
            bt.advance_host([LP_G], max_num_frames=1)
        s, c = bt.frame_tokens(0, f)
        assert dict(zip(s.tolist(), c.tolist())) == want, f
    assert not bool(bt.reached_final()[0])
    assert not bt.best_paths(True)[0].ok and not bt.best_paths(False)[0].ok
    # the lattice decoders: GetRawLattice returns false (a frame without tokens,
    # lattice-simple-decoder.cc:615-619); FinalRelativeCost is infinity (:562-572)
    lb = cabi.LatticeBatch(cabi.Graph(graph_g()), max_batch=1, beam=16.0, lattice_beam=4.0, min_active=0)
    lb.decode_host([LP_G])
    assert not lb.raw_lattices()[0].ok
    assert lb.final_relative_cost().tolist() == [INF]
    assert not bool(lb.reached_final()[0])


@pytest.mark.gpu
@pytest.mark.parametrize("chunk", [-1, 1, 2])
def test_gpu_prune_active_tokens_stops_early(chunk):
    from kaldi_decoder import cabi
    lb = cabi.LatticeBatch(cabi.Graph(graph_f()), max_batch=1, min_active=0, **OPTS_F)
    lb.init_decoding()
    if chunk < 0:
        lb.advance_host([LP_F])
    else:
        while lb.num_frames_decoded()[0] < 5:
            lb.advance_host([LP_F], max_num_frames=chunk)
    lat = lb.raw_lattices()[0]
    assert lat.arc_set() == BEFORE_F, chunk
    assert lat.final_set() == {(1, 0.5)}
    lb.finalize_decoding()
    assert lb.raw_lattices()[0].arc_set() == AFTER_F
    lb = cabi.LatticeBatch(cabi.Graph(graph_f()), max_batch=1, min_active=0,
                           **{**OPTS_F, "prune_scale": 0.1})
    lb.init_decoding()
    lb.advance_host([LP_F])
    assert lb.raw_lattices()[0].arc_set() == AFTER_F


@pytest.mark.gpu
@pytest.mark.parametrize("lattice_beam", [0.5, 0.375])
def test_gpu_final_pruning_needs_a_second_pass(lattice_beam):
    from kaldi_decoder import cabi
    arcs, finals = AFTER_H[lattice_beam]
    G = cabi.Graph(graph_h())
    lb = cabi.LatticeBatch(G, max_batch=1, beam=16.0, lattice_beam=lattice_beam, min_active=0)
    lb.init_decoding()
    lb.advance_host([LP_H])
    lat = lb.raw_lattices()[0]
    assert lat.arc_set() == set(ARCS_H.values())
    assert lat.final_set() == {(1, 0.0), (2, 2.0), (3, 0.0)}
    assert lb.final_relative_cost().tolist() == [0.0]
    lb.finalize_decoding()
    lat = lb.raw_lattices()[0]
    assert lat.arc_set() == arcs and lat.final_set() == finals
    assert lat.num_states == 1 + len(finals)
    # Decode(): the same in one launch
    lb = cabi.LatticeBatch(G, max_batch=1, beam=16.0, lattice_beam=lattice_beam, min_active=0)
    lb.decode_host([LP_H])
    lat = lb.raw_lattices()[0]
    assert lat.arc_set() == arcs and lat.final_set() == finals
