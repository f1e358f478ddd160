"""Known-answer tests worked out BY HAND from the reference source (not from the oracle).

The reference ships no decoder vectors (ref csrc/CMakeLists.txt:29-32) and cannot be
built here, so everything else in tests/ compares against oracle/ -- this repository's own
restatement.  These cases are the independent pin: graphs of at most seven states and two
frames, every weight a dyadic rational so that float and double arithmetic are exact, and
the expected tokens, cutoffs and paths derived in the comments from the cited lines of
/root/reference/kaldi-decoder/csrc/.  Both oracle policies (CPU) and the HIP path through
the C ABI (-m gpu) must reproduce them.
"""
import numpy as np
import pytest

from kaldi_decoder import synth

INF = float("inf")
INT_MAX = 2**31 - 1


# ---------------------------------------------------------------------------------------
# Graph A: an epsilon arc that improves a token which already exists.
#   state 0 (start)  a0: 1:10 /0.5   -> 1      a1: 2:20 /1.0  -> 2
#   state 1          a2: 0:30 /0.25  -> 3      a3: 1:0  /0.125 -> 1   (a2 is the epsilon arc)
#   state 2          a4: 1:0  /0     -> 3
#   state 3 (final 0.5)  a5: 2:40 /0.25 -> 3
# log-likelihoods (V = 2):  frame 0: [-1, -2]   frame 1: [-0.5, -4]
#
# FasterDecoder, beam 16, max_active INT_MAX, min_active 0:
#  InitDecoding (faster-decoder.cc:42-56): token (state 0, cost 0); state 0 has no
#    epsilon arc, ProcessNonemitting adds nothing.                    frontier {0: 0}
#  frame 0: GetCutoff fast path (:249-272): best 0, weight_cutoff 16, adaptive_beam 16.
#    new_weight = arc.weight + tok.cost + ac_cost (:210):
#      a0 -> 1: 0.5 + 0 + 1 = 1.5        a1 -> 2: 1.0 + 0 + 2 = 3.0
#    next_weight_cutoff ends at 1.5 + 16 = 17.5 (:215-217); both kept.
#    ProcessNonemitting(17.5) (:59-119): state 1 has a2 -> 3: 1.5 + 0.25 = 1.75.
#                                                       frontier {1: 1.5, 2: 3.0, 3: 1.75}
#  frame 1: best 1.5, weight_cutoff 17.5.
#      a3 from 1 -> 1: 0.125 + 1.5  + 0.5 = 2.125
#      a4 from 2 -> 3: 0     + 3.0  + 0.5 = 3.5
#      a5 from 3 -> 3: 0.25  + 1.75 + 4.0 = 6.0      (state 3 keeps 3.5, :220-228)
#    next cutoff 2.125 + 16.  ProcessNonemitting: a2 from 1 -> 3: 2.125 + 0.25 = 2.375,
#    strictly below 3.5: the existing token is replaced (:107-113).
#                                                       frontier {1: 2.125, 3: 2.375}
#  ReachedFinal (:347-354): state 3 is final -> true.  GetBestPath (:356-424): only final
#    states compete, best = 2.375 + 0.5; walk back: a2 <- a3 <- a0.
#    per-arc costs (:394-396): graph = arc weight, ac = float(cost - prev.cost) - graph:
#      a0: 1.5 - 0.5 = 1.0     a3: 0.625 - 0.125 = 0.5     a2: 0.25 - 0.25 = 0
#    final weight (0.5, 0) with use_final_probs, One() = (0, 0) without (:416-421).
def graph_a(finals=None):
    arcs = [(0, 1, 1, 10, 0.5), (0, 2, 2, 20, 1.0), (1, 3, 0, 30, 0.25), (1, 1, 1, 0, 0.125),
            (2, 3, 1, 0, 0.0), (3, 3, 2, 40, 0.25)]
    return synth.graph_from_arcs(4, 0, arcs, {3: 0.5} if finals is None else finals)


LP_A = np.array([[-1.0, -2.0], [-0.5, -4.0]], np.float32)
FRONTIERS_A = [{0: 0.0}, {1: 1.5, 2: 3.0, 3: 1.75}, {1: 2.125, 3: 2.375}]
PATH_A = dict(ilabel=[1, 1, 0], olabel=[10, 0, 30], state=[1, 1, 3], graph=[0.5, 0.125, 0.25],
              ac=[1.0, 0.5, 0.0])

# Graph A with no final state: ReachedFinal false, GetBestPath takes the cheapest token
# (:367-372) = state 1 at 2.125: a0, a3; the final weight is One() either way.
PATH_A_NOFINAL = dict(ilabel=[1, 1], olabel=[10, 0], state=[1, 1], graph=[0.5, 0.125], ac=[1.0, 0.5])

# ---------------------------------------------------------------------------------------
# Graph B: a three-way tie.  state 0 (start): a0 1:10/0.5, a1 1:11/0.5, a2 2:12/0.25, all
# -> state 1 (final 0).  frame 0: [-1, -1.25]: every arc costs exactly 1.5.  The first
# token inserted stays -- replacement needs a strictly lower cost (`*(e_found->val) <
# *new_tok`, faster-decoder.cc:220, Token::operator< is `cost_ > other.cost_`) -- and
# ArcIterator order is arc order: a0 wins, olabel 10.
def graph_b():
    return synth.graph_from_arcs(2, 0, [(0, 1, 1, 10, 0.5), (0, 1, 1, 11, 0.5), (0, 1, 2, 12, 0.25)],
                                 {1: 0.0})


LP_B = np.array([[-1.0, -1.25]], np.float32)
FRONTIERS_B = [{0: 0.0}, {1: 1.5}]
PATH_B = dict(ilabel=[1], olabel=[10], state=[1], graph=[0.5], ac=[1.0])

# ---------------------------------------------------------------------------------------
# Graph C: min_active binds.  beam 1, min_active 2, max_active INT_MAX, beam_delta 0.5.
#   state 0 (start)  a0 1:1/1.0 -> 1   a1 2:2/1.5 -> 2   a2 3:3/4.0 -> 3   a3 4:4/6.0 -> 4
#   state 1          a4 1:0/0 -> 1     a5 2:5/2.0 -> 5
#   state 2          a6 2:0/0 -> 2     a7 1:6/3.0 -> 6
#   state 3          a8 3:0/0 -> 3          state 4   a9 4:0/0 -> 4
#   state 5 final 0.25                      state 6   final 0
# log-likelihoods (V = 4):  frame 0: [0, 0, 0, 0]    frame 1: [0, -1, 0, 0]
#  frame 0: GetCutoff (:274-335) on one token: beam_cutoff 1; 1 > max_active no; 1 >
#    min_active (2) no -> min_active_cutoff stays +inf > beam_cutoff: adaptive_beam = inf -
#    0 + 0.5 = inf, weight_cutoff inf (:325-330).  Nothing is pruned:
#                                              frontier {1: 1.0, 2: 1.5, 3: 4.0, 4: 6.0}
#  frame 1: 4 tokens > min_active: tmp_array_ = [1, 1.5, 4, 6], nth_element at 2 ->
#    min_active_cutoff = 4.0 (:315-322) > beam_cutoff 2.0: adaptive_beam = 4 - 1 + 0.5 = 3.5,
#    weight_cutoff 4.0.  Expanded: cost < 4.0 (:202, strict) = states 1 and 2 only.
#    best token (state 1, cost 1.0) first (:176-189): a4 -> 1.0, a5 -> 2 + 1 + 1 = 4.0;
#    next_weight_cutoff = 1.0 + 3.5 = 4.5.
#      a4 -> 1: 0 + 1.0 + 0 = 1.0      a5 -> 5: 2.0 + 1.0 + 1.0 = 4.0  (< 4.5: kept)
#      a6 -> 2: 0 + 1.5 + 1 = 2.5      a7 -> 6: 3.0 + 1.5 + 0 = 4.5  (not < 4.5: pruned, :211)
#                                              frontier {1: 1.0, 2: 2.5, 5: 4.0}
#  ReachedFinal: state 5.  GetBestPath limits itself to final states whenever one was
#    reached, whatever use_final_probs says (:362-384): token 5, path a0, a5;
#    a0: graph 1.0, ac 0     a5: graph 2.0, ac float(4.0 - 1.0) - 2.0 = 1.0
#    final weight (0.25, 0) / One().
def graph_c():
    arcs = [(0, 1, 1, 1, 1.0), (0, 2, 2, 2, 1.5), (0, 3, 3, 3, 4.0), (0, 4, 4, 4, 6.0),
            (1, 1, 1, 0, 0.0), (1, 5, 2, 5, 2.0), (2, 2, 2, 0, 0.0), (2, 6, 1, 6, 3.0),
            (3, 3, 3, 0, 0.0), (4, 4, 4, 0, 0.0)]
    return synth.graph_from_arcs(7, 0, arcs, {5: 0.25, 6: 0.0})


LP_C = np.array([[0, 0, 0, 0], [0, -1, 0, 0]], np.float32)
OPTS_C = dict(beam=1.0, min_active=2, max_active=INT_MAX, beam_delta=0.5)
FRONTIERS_C = [{0: 0.0}, {1: 1.0, 2: 1.5, 3: 4.0, 4: 6.0}, {1: 1.0, 2: 2.5, 5: 4.0}]
PATH_C = dict(ilabel=[1, 2], olabel=[1, 5], state=[1, 5], graph=[1.0, 2.0], ac=[0.0, 1.0])

# ---------------------------------------------------------------------------------------
# Graph C again, max_active binding: beam 10, max_active 2, min_active 0, beam_delta 0.5.
#  frame 0: one token: not more than max_active (:289), more than min_active = 0 ->
#    min_active_cutoff = best_cost = 0 (:312-313), not above beam_cutoff 10: adaptive_beam =
#    beam = 10, weight_cutoff 10 (:331-334).  new weights 1.0, 1.5, 4.0, 6.0; the running
#    cutoff ends at 1.0 + 10 = 11: all kept.          frontier {1: 1.0, 2: 1.5, 3: 4.0, 4: 6.0}
#  frame 1: 4 tokens > max_active: nth_element at 2 of [1, 1.5, 4, 6] -> max_active_cutoff =
#    4.0 (:291-295) < beam_cutoff 11: adaptive_beam = 4.0 - 1.0 + 0.5 = 3.5, weight_cutoff 4.0
#    (:297-303).  Expanded: cost < 4.0 = states 1 and 2 -- max_active tokens.  From here on
#    exactly as in the min_active case above (same weight_cutoff 4.0, same adaptive_beam 3.5):
#                                                     frontier {1: 1.0, 2: 2.5, 5: 4.0}
OPTS_C_MAX = dict(beam=10.0, min_active=0, max_active=2, beam_delta=0.5)

FASTER_CASES = [
    ("eps-improves", graph_a, LP_A, dict(beam=16.0, min_active=0, max_active=INT_MAX), FRONTIERS_A,
     True, PATH_A, (0.5, 0.0)),
    ("no-final", lambda: graph_a({}), LP_A, dict(beam=16.0, min_active=0, max_active=INT_MAX),
     FRONTIERS_A, False, PATH_A_NOFINAL, (0.0, 0.0)),
    ("tie", graph_b, LP_B, dict(beam=16.0, min_active=0, max_active=INT_MAX), FRONTIERS_B, True,
     PATH_B, (0.0, 0.0)),
    ("min-active", graph_c, LP_C, OPTS_C, FRONTIERS_C, True, PATH_C, (0.25, 0.0)),
    ("max-active", graph_c, LP_C, OPTS_C_MAX, FRONTIERS_C, True, PATH_C, (0.25, 0.0)),
]

# ---------------------------------------------------------------------------------------
# Graph A with input labels that do not fit 20 bits (round 6).  The reference's loop takes any
# int32 `arc.ilabel` (faster-decoder.cc:204-209); the GPU's arc record packs the label into 20
# bits, whose largest value is an escape code into a side array of full labels
# (kd_device.h kIlabelEscape).  Label 1 becomes 2^20 - 1 (the escape value itself: it must
# go through the side array too), label 2 becomes 2^20 + 5; the matrix has 2^20 + 5 columns,
# the two that matter holding graph A's log-likelihoods.  Tokens, costs and paths are graph
# A's, the derivation above word for word; only the path's input labels change.
BIG1, BIG2 = (1 << 20) - 1, (1 << 20) + 5


def graph_a_big(finals=None):
    relabel = {0: 0, 1: BIG1, 2: BIG2}
    arcs = [(0, 1, 1, 10, 0.5), (0, 2, 2, 20, 1.0), (1, 3, 0, 30, 0.25), (1, 1, 1, 0, 0.125),
            (2, 3, 1, 0, 0.0), (3, 3, 2, 40, 0.25)]
    return synth.graph_from_arcs(4, 0, [(s, d, relabel[i], o, w) for s, d, i, o, w in arcs],
                                 {3: 0.5} if finals is None else finals)


def lp_a_big():
    lp = np.full((2, BIG2), -100.0, np.float32)
    lp[:, BIG1 - 1] = LP_A[:, 0]
    lp[:, BIG2 - 1] = LP_A[:, 1]
    return lp


PATH_A_BIG = dict(PATH_A, ilabel=[BIG1, BIG1, 0])
LAT_ARCS_BIG = {k: v[:4] + ({0: 0, 1: BIG1, 2: BIG2}[v[4]],) + v[5:] for k, v in {
    "L0": (0, 0, 1, 1, 1, 10, 0.5, 1.0), "L1": (0, 0, 1, 2, 2, 20, 1.0, 2.0),
    "L2": (1, 1, 1, 3, 0, 30, 0.25, 0.0), "L3": (1, 1, 2, 1, 1, 0, 0.125, 0.5),
    "L4": (1, 2, 2, 3, 1, 0, 0.0, 0.5), "L5": (1, 3, 2, 3, 2, 40, 0.25, 4.0),
    "L6": (2, 1, 2, 3, 0, 30, 0.25, 0.0)}.items()}

# ---------------------------------------------------------------------------------------
# LatticeSimpleDecoder on graph A, beam 16 (float costs, tot = cur + ac + graph,
# lattice-simple-decoder.cc:383-385; epsilon tot = cur + graph, :170-172).  The tokens are
# the FasterDecoder's above; every scored arc below the cutoff is a ForwardLink (:393-398,
# :179-180) -- including a5 into state 3 at frame 1 although that token already costs less:
#   frame 0 tokens {0: 0}; frame 1 {1: 1.5, 2: 3.0, 3: 1.75}; frame 2 {1: 2.125, 3: 2.375}
#   links  L0 (0,0)->(1,1) a0   L1 (0,0)->(1,2) a1   L2 (1,1)->(1,3) a2 (epsilon)
#          L3 (1,1)->(2,1) a3   L4 (1,2)->(2,3) a4   L5 (1,3)->(2,3) a5
#          L6 (2,1)->(2,3) a2 (epsilon; regenerated from state 1's final cost, :164-165)
# FinalizeDecoding (:407-520): final costs: state 3: 2.375 + 0.5 = 2.875 (state 1 is not
#   final); FinalRelativeCost = 2.875 - 2.125 = 0.75 (:562-572).  Last frame extras
#   (:467-509): token 3: 0; token 1: own term inf, through L6: 0 + (2.125 + 0.25 - 2.375) = 0.
#   link_extra_cost = next.extra + (tok.tot + ac + graph - next.tot) (:257-260):
#     L3: 0 + (1.5 + 0.5 + 0.125 - 2.125) = 0        L4: 0 + (3.0 + 0.5 + 0 - 2.375) = 1.125
#     L5: 0 + (1.75 + 4 + 0.25 - 2.375)   = 3.625    => extras at frame 1: {1: 0, 2: 1.125, 3: 3.625}
#     L2: 3.625 + (1.5 + 0.25 - 1.75) = 3.625 (token 1 keeps min(0, 3.625) = 0)
#     L0: 0 + 0 = 0      L1: 1.125 + (0 + 2 + 1 - 3) = 1.125
#   lattice_beam 10: all seven links survive.  lattice_beam 2: L5 (3.625) goes, token (1,3)
#   is left without links -> extra inf -> freed (:294-295, :320-329), and L2 with it: five arcs.
#   Final state: (2,3) with weight 0.5 (:645-652).
LAT_ARCS = {
    "L0": (0, 0, 1, 1, 1, 10, 0.5, 1.0), "L1": (0, 0, 1, 2, 2, 20, 1.0, 2.0),
    "L2": (1, 1, 1, 3, 0, 30, 0.25, 0.0), "L3": (1, 1, 2, 1, 1, 0, 0.125, 0.5),
    "L4": (1, 2, 2, 3, 1, 0, 0.0, 0.5), "L5": (1, 3, 2, 3, 2, 40, 0.25, 4.0),
    "L6": (2, 1, 2, 3, 0, 30, 0.25, 0.0),
}
LATTICE_CASES = [(10.0, ["L0", "L1", "L2", "L3", "L4", "L5", "L6"]),
                 (2.0, ["L0", "L1", "L3", "L4", "L6"])]


def _check_path(p, want, fw, tag):
    assert p.ok, tag
    assert p.ilabel.tolist() == want["ilabel"], tag
    assert p.olabel.tolist() == want["olabel"], tag
    assert p.state.tolist() == want["state"], tag
    assert p.graph_cost.tolist() == want["graph"], tag
    assert p.ac_cost.tolist() == want["ac"], tag
    assert tuple(float(x) for x in p.final_weight) == fw, tag


@pytest.mark.parametrize("policy", [0, 1])
@pytest.mark.parametrize("case", FASTER_CASES, ids=[c[0] for c in FASTER_CASES])
def test_oracle_faster_decoder_kat(oracle, case, policy):
    name, mk, lp, opts, frontiers, reached, path, fw = case
    d = oracle.FasterDecoder(oracle.Graph(mk()), policy=policy, **opts)
    d.init_decoding()
    for f, want in enumerate(frontiers):
        if f:
            d.advance_decoding(lp, max_num_frames=1)
        s, c = d.tokens()
        assert dict(zip(s.tolist(), c.tolist())) == want, (name, policy, f)
    assert d.reached_final() == reached
    _check_path(d.get_best_path(True), path, fw, (name, policy))
    _check_path(d.get_best_path(False), path, (0.0, 0.0), (name, policy))


@pytest.mark.parametrize("lattice_beam,want", LATTICE_CASES)
def test_oracle_lattice_decoders_kat(oracle, lattice_beam, want):
    G = oracle.Graph(graph_a())
    arcs = {LAT_ARCS[k] for k in want}
    for dec in (oracle.LatticeSimpleDecoder(G, beam=16.0, lattice_beam=lattice_beam, prune_interval=1),
                oracle.LatticeSimpleDecoder(G, beam=16.0, lattice_beam=lattice_beam),
                oracle.CanonLatticeDecoder(G, beam=16.0, lattice_beam=lattice_beam)):
        assert dec.decode(LP_A)
        assert dec.final_relative_cost() == 0.75
        lat = dec.get_raw_lattice()
        assert lat.arc_set() == arcs
        assert lat.final_set() == {(3, 0.5)}
        p = dec.get_best_path()
        assert p.isyms.tolist() == [1, 1] and p.osyms.tolist() == [10, 30]


def test_oracle_simple_decoder_kat(oracle):
    # SimpleDecoder (simple-decoder.cc:150-281) with a beam this wide prunes nothing: the
    # same tokens and the same path as the FasterDecoder on graph A; its arcs carry the
    # arc's own (graph, acoustic) pair (:124-133).
    for cls in (oracle.SimpleDecoder, oracle.CanonSimpleDecoder):
        d = cls(oracle.Graph(graph_a()), beam=16.0)
        d.decode(LP_A)
        assert d.reached_final()
        _check_path(d.get_best_path(True), PATH_A, (0.5, 0.0), cls.__name__)
        assert d.final_relative_cost() == 0.75  # (2.375 + 0.5) - 2.125, :78-100


@pytest.mark.gpu
@pytest.mark.parametrize("case", FASTER_CASES, ids=[c[0] for c in FASTER_CASES])
def test_gpu_faster_decoder_kat(case):
    from kaldi_decoder import cabi
    name, mk, lp, opts, frontiers, reached, path, fw = case
    bt = cabi.Batch(cabi.Graph(mk()), max_batch=1, **opts)
    bt.init_decoding()
    for f, want in enumerate(frontiers):
        if f:
            bt.advance_host([lp], max_num_frames=1)
        s, c = bt.frame_tokens(0, f)
        assert dict(zip(s.tolist(), c.tolist())) == want, (name, f)
    assert bool(bt.reached_final()[0]) == reached
    _check_path(bt.best_paths(True)[0], path, fw, name)
    _check_path(bt.best_paths(False)[0], path, (0.0, 0.0), name)


@pytest.mark.gpu
@pytest.mark.parametrize("lattice_beam,want", LATTICE_CASES)
def test_gpu_lattice_decoder_kat(lattice_beam, want):
    from kaldi_decoder import cabi
    lb = cabi.LatticeBatch(cabi.Graph(graph_a()), max_batch=1, beam=16.0, lattice_beam=lattice_beam,
                           min_active=0)
    lb.init_decoding()
    lb.advance_host([LP_A])
    assert lb.raw_lattices()[0].arc_set() == set(LAT_ARCS.values())  # before any pruning
    lb.finalize_decoding()
    assert lb.final_relative_cost().tolist() == [0.75]
    lat = lb.raw_lattices()[0]
    assert lat.arc_set() == {LAT_ARCS[k] for k in want}
    assert lat.final_set() == {(3, 0.5)}


@pytest.mark.parametrize("policy", [0, 1])
def test_oracle_input_labels_beyond_20_bits(oracle, policy):
    lp = lp_a_big()
    d = oracle.FasterDecoder(oracle.Graph(graph_a_big()), policy=policy, beam=16.0, min_active=0)
    d.decode(lp)
    s, c = d.tokens()
    assert dict(zip(s.tolist(), c.tolist())) == FRONTIERS_A[2]
    _check_path(d.get_best_path(True), PATH_A_BIG, (0.5, 0.0), policy)


@pytest.mark.gpu
def test_gpu_input_labels_beyond_20_bits():
    """kd_graph_create refuses nothing the reference decodes except epsilon cycles of total
    weight zero or less: labels of 2^20 - 1 and more."""
    from kaldi_decoder import cabi
    lp = lp_a_big()
    G = cabi.Graph(graph_a_big())
    bt = cabi.Batch(G, max_batch=1, beam=16.0, min_active=0)
    bt.init_decoding()
    for f, want in enumerate(FRONTIERS_A):
        if f:
            bt.advance_host([lp], max_num_frames=1)
        s, c = bt.frame_tokens(0, f)
        assert dict(zip(s.tolist(), c.tolist())) == want, f
    assert bool(bt.reached_final()[0])
    _check_path(bt.best_paths(True)[0], PATH_A_BIG, (0.5, 0.0), "big")
    for lattice_beam, want in LATTICE_CASES:
        lb = cabi.LatticeBatch(G, max_batch=1, beam=16.0, lattice_beam=lattice_beam, min_active=0)
        lb.init_decoding()
        lb.advance_host([lp])
        assert lb.raw_lattices()[0].arc_set() == set(LAT_ARCS_BIG.values())
        lb.finalize_decoding()
        assert lb.final_relative_cost().tolist() == [0.75]
        assert lb.raw_lattices()[0].arc_set() == {LAT_ARCS_BIG[k] for k in want}
    sb = cabi.SimpleBatch(G, max_batch=1, beam=16.0)
    sb.init_decoding()
    sb.advance_host([lp])
    _check_path(sb.best_paths(True)[0], PATH_A_BIG, (0.5, 0.0), "simple")
    # a matrix one column short of the largest label is refused, as for any graph
    with pytest.raises(RuntimeError, match="largest ilabel"):
        bt.init_decoding()
        bt.advance_host([lp[:, :BIG2 - 1].copy()])


@pytest.mark.gpu
def test_gpu_simple_decoder_kat():
    from kaldi_decoder import cabi
    sb = cabi.SimpleBatch(cabi.Graph(graph_a()), max_batch=1, beam=16.0)
    sb.init_decoding()
    sb.advance_host([LP_A])
    assert bool(sb.reached_final()[0])
    _check_path(sb.best_paths(True)[0], PATH_A, (0.5, 0.0), "simple")
    assert sb.final_relative_cost().tolist() == [0.75]
