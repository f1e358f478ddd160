"""The HIP path (through the C ABI) against the committed golden vectors."""
import numpy as np
import pytest

import golden_util as gu
from kaldi_decoder import cabi

pytestmark = pytest.mark.gpu


def _best_path_syms(lat):
    """Tropical shortest path through a cabi.RawLattice (Bellman-Ford over the arcs; the
    lattice is acyclic), epsilon-free label sequences."""
    n = lat.num_states
    cost = np.full(n, np.inf)
    cost[lat.start] = 0.0
    w = lat.graph_cost.astype(np.float64) + lat.ac_cost.astype(np.float64)
    back = np.full(n, -1, np.int64)
    for _ in range(n + 1):
        cand = cost[lat.src] + w
        better = cand < cost[lat.dst]
        if not better.any():
            break
        for a in np.nonzero(better)[0]:
            if cand[a] < cost[lat.dst[a]]:
                cost[lat.dst[a]] = cand[a]
                back[lat.dst[a]] = a
    fin = cost[lat.final_state] + lat.final_graph
    if not len(fin) or not np.isfinite(fin).any():
        return False, [], []
    s = int(lat.final_state[int(np.argmin(fin))])
    il, ol = [], []
    while s != lat.start:
        a = int(back[s])
        il.append(int(lat.ilabel[a]))
        ol.append(int(lat.olabel[a]))
        s = int(lat.src[a])
    return True, [x for x in il[::-1] if x], [x for x in ol[::-1] if x]


@pytest.mark.parametrize("name", gu.names("g1_") + gu.names("g2_") + gu.names("g3_") + gu.names("g5_"))
def test_faster_decoder_matches_golden(name):
    z, g, opts = gu.load(name)
    lp = z["logprobs"]
    T = lp.shape[0]
    bt = cabi.Batch(cabi.Graph(g), max_batch=1, **opts)
    # G4: the same input in chunks of 7 frames
    bt.init_decoding()
    counts = [len(bt.frame_tokens(0, 0)[0])]
    for t0 in range(0, T, 7):
        bt.advance_host([lp], max_num_frames=7)
    assert bt.num_frames_decoded().tolist() == [int(z["canon_num_frames_decoded"])]
    counts += [len(bt.frame_tokens(0, f)[0]) for f in range(1, T + 1)]
    assert counts == z["canon_tokens_per_frame"].tolist(), name
    s, c = bt.frame_tokens(0, T)
    assert s.tolist() == z["canon_last_states"].tolist() and c.tolist() == z["canon_last_costs"].tolist()
    assert bool(bt.reached_final()[0]) == bool(z["canon_reached_final"])
    for uf in (1, 0):
        gu.assert_path(bt.best_paths(bool(uf))[0], z, f"canon_uf{uf}_", (name, uf))
        # the reference-order answer (policy 0 = faster-decoder.cc:155-241 as written, hash
        # order and running cutoff included): the same best path, arc for arc
        gu.assert_path(bt.best_paths(bool(uf))[0], z, f"ref_uf{uf}_", (name, uf, "reference order"))
    assert bool(bt.reached_final()[0]) == bool(z["ref_reached_final"])
    assert bt.num_frames_decoded().tolist() == [int(z["ref_num_frames_decoded"])]


@pytest.mark.parametrize("name", gu.names("g6_"))
def test_lattice_decoder_matches_golden(name):
    z, g, opts = gu.load(name)
    lb = cabi.LatticeBatch(cabi.Graph(g), max_batch=1, **opts)
    lb.init_decoding()
    lb.advance_host([z["logprobs"]])
    created = gu.lattice_rows(lb.raw_lattices()[0])  # every forward link, before any pruning
    lb.finalize_decoding()
    lat = lb.raw_lattices()[0]
    assert np.array_equal(gu.lattice_rows(lat), z["canon_arcs"]), name
    assert lat.num_states == int(z["canon_num_states"])
    assert np.float32(lb.final_relative_cost()[0]) == z["canon_final_relative_cost"]
    ref_rows = {tuple(r) for r in z["ref_arcs"].tolist()}
    assert all(tuple(r) in ref_rows for r in gu.lattice_rows(lat).tolist())
    assert all(tuple(r) in {tuple(q) for q in z["ref_pi1_arcs"].tolist()}
               for r in gu.lattice_rows(lat).tolist())
    # every reference-order arc is below (frame best + beam) iff the GPU search created
    # that link too; the reference's other arcs are the order-dependent extras
    gu.assert_extras_beyond_cutoff(z["ref_arcs"], created, z["canon_frame_best"], opts["beam"],
                                   g["start"], name)
    # 1-best through the GPU lattice == 1-best of the reference-order decoder
    ok, isyms, osyms = _best_path_syms(lat)
    assert ok and isyms == z["ref_isyms"].tolist() and osyms == z["ref_osyms"].tolist()
