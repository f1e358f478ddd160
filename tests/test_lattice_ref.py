"""The oracle's lattice decoders against an INDEPENDENT definition of the finalized raw lattice
(tests/lattice_ref.py: exact forward / backward Viterbi in float64, no restatement of the
reference's pruning recursion).  With a search beam that prunes nothing, LatticeSimpleDecoder
in the reference's order (policy 0) and the canonical decoder must both produce exactly the
links and tokens within lattice_beam of the best path -- on acyclic graphs and on graphs with
epsilon cycles of positive weight.  tests/test_gpu_lattice.py runs the same check on the GPU."""
import numpy as np
import pytest

from kaldi_decoder import synth
import lattice_ref

WIDE = 1.0e6  # search beam: nothing is pruned during the search


def _cases():
    # (name, graph, lexicon, V, T, lattice beams)
    for seed in range(3):
        g = synth.random_graph(60 + 40 * seed, 3.0, 10, 0.25, 100 + seed)
        yield f"rand{seed}", g, None, 10, 12, (2.0, 5.0)
    for seed in range(2):
        g = synth.random_graph(80, 3.5, 10, 0.45, 201 + seed, eps_back=0.6)
        yield f"cyclic{seed}", g, None, 10, 10, (1.5, 4.0)
    g = synth.hlg_graph(V=12, n_words=30, n_hist=30, avg_succ=4, seed=5)
    yield "HLG", g, g["lexicon"], 12, 14, (3.0, 8.0)
    g = synth.hl_graph(V=12, n_words=40, seed=6)
    yield "HL", g, g["lexicon"], 12, 14, (3.0,)


CASES = list(_cases())


def _has_eps_cycle(g):
    import collections
    S = g["num_states"]
    indeg = np.zeros(S, np.int64)
    src = np.repeat(np.arange(S), np.diff(g["row_ptr"]))
    eps = g["ilabel"] == 0
    np.add.at(indeg, g["nextstate"][eps], 1)
    out = collections.defaultdict(list)
    for a, b in zip(src[eps].tolist(), g["nextstate"][eps].tolist()):
        out[a].append(b)
    todo = [s for s in range(S) if indeg[s] == 0]
    n = 0
    while todo:
        s = todo.pop()
        n += 1
        for d in out[s]:
            indeg[d] -= 1
            if indeg[d] == 0:
                todo.append(d)
    return n < S


def test_the_cyclic_cases_do_hold_epsilon_cycles():
    for name, g, *_ in CASES:
        assert _has_eps_cycle(g) == name.startswith("cyclic"), name


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_oracle_lattices_are_the_links_within_lattice_beam_of_the_best_path(oracle, case):
    name, g, lex, V, T, lbeams = case
    og = oracle.Graph(g)
    margin = 1e-4 * (T + 1)  # float32 costs summed over T frames against float64
    for utt in range(3):
        lp = synth.make_logprobs(T, V, 300 + utt, peak=2.0, lexicon=lex)
        for lb in lbeams:
            d0 = oracle.LatticeSimpleDecoder(og, beam=WIDE, lattice_beam=lb)
            d0.decode(lp)
            d1 = oracle.CanonLatticeDecoder(og, beam=WIDE, lattice_beam=lb)
            d1.decode(lp)
            l0, l1 = d0.get_raw_lattice(), d1.get_raw_lattice()
            n_margin = lattice_ref.check_lattice(l0, g, lp, lb, margin)
            assert lattice_ref.check_lattice(l1, g, lp, lb, margin) == n_margin
            # (the check must have teeth: hardly anything may hide inside the margin)
            assert n_margin <= max(2, len(l0.src) // 50), (name, n_margin, len(l0.src))
            # with nothing pruned by the search the two orders give one lattice
            assert l0.arc_set() == l1.arc_set() and l0.state_set() == l1.state_set()


def test_the_check_rejects_a_lattice_with_a_link_too_few_or_too_many(oracle):
    g = synth.random_graph(60, 3.0, 10, 0.25, 100)
    og = oracle.Graph(g)
    lp = synth.make_logprobs(12, 10, 300, peak=2.0)
    d = oracle.CanonLatticeDecoder(og, beam=WIDE, lattice_beam=3.0)
    d.decode(lp)
    lat = d.get_raw_lattice()
    links, _, _ = lattice_ref.lattice_by_viterbi(g, lp, 3.0)
    lattice_ref.check_lattice(lat, g, lp, 3.0, 1e-3)
    # one arc fewer: the arc with the smallest extra cost is certainly outside the margin
    f, s = lat.state_frame, lat.state_graph
    keys = list(zip(f[lat.src].tolist(), s[lat.src].tolist(), f[lat.dst].tolist(), s[lat.dst].tolist(),
                    lat.ilabel.tolist(), lat.olabel.tolist()))
    k = int(np.argmin([links[key] for key in keys]))
    keep = np.arange(len(lat.src)) != k
    import copy
    fewer = copy.copy(lat)
    for name in ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost"):
        setattr(fewer, name, getattr(lat, name)[keep])
    with pytest.raises(AssertionError):
        lattice_ref.check_lattice(fewer, g, lp, 3.0, 1e-3)
    # a wider lattice checked against a narrower beam holds links that must not be there
    d = oracle.CanonLatticeDecoder(og, beam=WIDE, lattice_beam=6.0)
    d.decode(lp)
    with pytest.raises(AssertionError):
        lattice_ref.check_lattice(d.get_raw_lattice(), g, lp, 3.0, 1e-3)
