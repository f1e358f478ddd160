"""The HIP path (through the C ABI) against the committed golden vectors."""
import numpy as np
import pytest

import golden_util as gu
from kaldi_decoder import cabi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.names("g1_") + gu.names("g3_") + gu.names("g5_"))
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


@pytest.mark.parametrize("name", gu.names("g6_"))
def test_lattice_decoder_matches_golden(name):
    z, g, opts = gu.load(name)
    lb = cabi.LatticeBatch(cabi.Graph(g), max_batch=1, **opts)
    lb.init_decoding()
    lb.advance_host([z["logprobs"]])
    lb.finalize_decoding()
    lat = lb.raw_lattices()[0]
    assert np.array_equal(gu.lattice_rows(lat), z["canon_arcs"]), name
    assert lat.num_states == int(z["canon_num_states"])
    assert np.float32(lb.final_relative_cost()[0]) == z["canon_final_relative_cost"]
    ref_rows = {tuple(r) for r in z["ref_arcs"].tolist()}
    assert all(tuple(r) in ref_rows for r in gu.lattice_rows(lat).tolist())
