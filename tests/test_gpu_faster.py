"""GPU parity: the HIP FasterDecoder (through the C ABI) against the oracle's
canonical policy -- bit-exact on states/labels and on every cost, which are the
same IEEE expressions on both sides (integer work: identical; floating point:
the north star allows 1e-4, we assert equality and report the max deviation)."""
import numpy as np
import pytest

from kaldi_decoder import cabi, synth

pytestmark = pytest.mark.gpu


def _V(g):
    return int(g["ilabel"].max())


def _cases():
    yield "H8", synth.ctc_topo(8), None, dict(beam=6.0)
    yield "H32", synth.ctc_topo(32), None, dict(beam=16.0)
    yield "H500", synth.ctc_topo(500), None, dict(beam=15.0)
    for seed in range(3):
        yield (f"rand{seed}", synth.random_graph(200 + 300 * seed, 3.0, 24, 0.25, seed), None,
               dict(beam=6.0 + 3 * seed, min_active=0))
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    yield "HL", g, g["lexicon"], dict(beam=12.0, max_active=300)
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    yield "HLG", g, g["lexicon"], dict(beam=10.0)
    yield "HLGmin", g, g["lexicon"], dict(beam=4.0, min_active=50)
    yield "HLGmax", g, g["lexicon"], dict(beam=12.0, max_active=40, min_active=0)


def _assert_same_path(pg, po, tag):
    assert pg.ok == po.ok, tag
    assert pg.ilabel.tolist() == po.ilabel.tolist(), tag
    assert pg.olabel.tolist() == po.olabel.tolist(), tag
    assert pg.state.tolist() == po.state.tolist(), tag
    assert pg.graph_cost.tolist() == po.graph_cost.tolist(), tag
    assert pg.ac_cost.tolist() == po.ac_cost.tolist(), tag
    assert pg.final_weight == po.final_weight, tag


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_frontier_and_best_path_match_oracle(oracle, case):
    name, g, lex, opts = case
    T, B = 40, 3
    G = cabi.Graph(g)
    OG = oracle.Graph(g)
    lps = [synth.make_logprobs(T - 5 * i, _V(g), i, peak=6.0, lexicon=lex) for i in range(B)]
    bt = cabi.Batch(G, max_batch=B, **opts)
    bt.init_decoding()
    decs = [oracle.FasterDecoder(OG, policy=1, **opts) for _ in range(B)]
    for d in decs:
        d.init_decoding()
    for i in range(B):
        s, c = bt.frame_tokens(i, 0)
        so, co = decs[i].tokens()
        assert s.tolist() == so.tolist() and c.tolist() == co.tolist(), (name, "init")
    step = 7
    for t0 in range(0, T, step):
        bt.advance_host(lps, max_num_frames=step)
        for i, d in enumerate(decs):
            d.advance_decoding(lps[i], max_num_frames=step)
            n = d.num_frames_decoded()
            s, c = bt.frame_tokens(i, n)
            so, co = d.tokens()
            assert s.tolist() == so.tolist(), (name, i, n)
            assert c.tolist() == co.tolist(), (name, i, n)
    assert bt.num_frames_decoded().tolist() == [lp.shape[0] for lp in lps]
    rf = bt.reached_final()
    for use_final in (True, False):
        paths = bt.best_paths(use_final)
        for i, d in enumerate(decs):
            assert bool(rf[i]) == d.reached_final()
            _assert_same_path(paths[i], d.get_best_path(use_final), (name, i, use_final))
    st = bt.stats()
    for i, d in enumerate(decs):
        so = d.stats()
        assert st[i, 0] == so["frames"]
        assert st[i, 1] == so["tokens_in"]
        assert st[i, 2] == so["tokens_expanded"]
        assert st[i, 3] == so["arcs_scanned"]
        assert st[i, 5] == so["tokens_out"]
        # epsilon re-expansion counts depend on relaxation order (worklist rounds
        # here, a LIFO queue in the reference): not a parity quantity
        assert st[i, 6] > 0 or so["eps_arcs_scanned"] == 0


def test_every_frame_of_one_utterance(oracle):
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(60, 40, 3, peak=5.0, lexicon=g["lexicon"])
    bt = cabi.Batch(G, max_batch=1, beam=9.0)
    bt.init_decoding()
    bt.advance_host([lp])
    d = oracle.FasterDecoder(OG, beam=9.0, policy=1)
    d.init_decoding()
    for t in range(60):
        d.advance_decoding(lp, max_num_frames=1)
        s, c = bt.frame_tokens(0, t + 1)
        so, co = d.tokens()
        assert s.tolist() == so.tolist() and c.tolist() == co.tolist(), t


def test_spill_to_global_table(oracle):
    # a frontier far beyond the 8192-slot LDS table: dense H with a flat posterior
    V = 500
    g = synth.random_graph(30000, 4.0, 16, 0.1, 9, max_weight=0.5)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(12, 16, 0, peak=0.0)
    opts = dict(beam=1000.0, min_active=0)
    bt = cabi.Batch(G, max_batch=1, **opts)
    bt.init_decoding()
    bt.advance_host([lp])
    d = oracle.FasterDecoder(OG, policy=1, **opts)
    d.decode(lp)
    s, c = bt.frame_tokens(0, 12)
    so, co = d.tokens()
    assert len(so) > 8192
    assert s.tolist() == so.tolist() and c.tolist() == co.tolist()
    st = dict(zip(cabi.STAT_NAMES, bt.stats()[0].tolist()))
    # all three tiers ran: LDS-resident small frames, the big LDS table, the global table
    assert st["spill_frames"] > 0 and st["small_frames"] > 0
    _assert_same_path(bt.best_paths()[0], d.get_best_path(), "spill")


def test_small_frame_outgrows_its_table(oracle):
    # a frontier of at most 1024 tokens whose successors do not fit the 4096-slot table of the
    # LDS-resident tier: its candidate queue is replayed in the big table
    g = synth.random_graph(30000, 8.0, 16, 0.1, 11, max_weight=0.5)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(5, 16, 0, peak=0.0)
    opts = dict(beam=1000.0, min_active=0)
    bt = cabi.Batch(G, max_batch=1, **opts)
    bt.init_decoding()
    bt.advance_host([lp])
    d = oracle.FasterDecoder(OG, policy=1, **opts)
    d.decode(lp)
    n = [len(bt.frame_tokens(0, f)[0]) for f in range(6)]
    assert any(n[f] <= 1024 and n[f + 1] > 3072 for f in range(5)), n
    s, c = bt.frame_tokens(0, 5)
    so, co = d.tokens()
    assert s.tolist() == so.tolist() and c.tolist() == co.tolist()
    _assert_same_path(bt.best_paths()[0], d.get_best_path(), "outgrown")


def test_big_lds_table_tier(oracle):
    # frontiers between 1024 and 6144 tokens run in the 8192-slot LDS table; V > 2048 also
    # takes the log-likelihood row from global memory instead of LDS
    g = synth.random_graph(3000, 3.0, 3000, 0.1, 5, max_weight=0.5)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(10, 3000, 0, peak=0.0)
    opts = dict(beam=1000.0, min_active=0)
    bt = cabi.Batch(G, max_batch=1, **opts)
    bt.init_decoding()
    bt.advance_host([lp])
    d = oracle.FasterDecoder(OG, policy=1, **opts)
    d.decode(lp)
    s, c = bt.frame_tokens(0, 10)
    so, co = d.tokens()
    assert 1024 < len(so) <= 6144
    assert s.tolist() == so.tolist() and c.tolist() == co.tolist()
    st = dict(zip(cabi.STAT_NAMES, bt.stats()[0].tolist()))
    assert st["frames"] - st["small_frames"] - st["spill_frames"] > 0, st
    _assert_same_path(bt.best_paths()[0], d.get_best_path(), "big table")


def test_streaming_with_decodable_offset(oracle):
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    G = cabi.Graph(g)
    lp = synth.make_logprobs(50, 40, 1, peak=6.0, lexicon=g["lexicon"])
    a = cabi.Batch(G, max_batch=1, beam=8.0)
    a.init_decoding()
    a.advance_host([lp])
    b = cabi.Batch(G, max_batch=1, beam=8.0)
    b.init_decoding()
    for start in range(0, 50, 9):  # DecodableCtc(chunk, offset=start)
        b.advance_host([lp[start:start + 9]], offsets=[start])
    assert b.num_frames_decoded().tolist() == [50]
    _assert_same_path(a.best_paths()[0], b.best_paths()[0], "stream")


def test_device_pointers_and_reinit(oracle):
    import torch
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T = 4, 30
    lps = [synth.make_logprobs(T, 40, i, peak=6.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    bt = cabi.Batch(G, max_batch=B, beam=8.0)
    for _ in range(2):  # InitDecoding can be called again for a new utterance
        bt.init_decoding()
        bt.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, 40)
        paths = bt.best_paths()
        for i in range(B):
            d = oracle.FasterDecoder(OG, beam=8.0, policy=1)
            d.decode(lps[i])
            _assert_same_path(paths[i], d.get_best_path(), i)
    assert bt.last_kernel_ms() > 0


def test_error_behaviour():
    g = synth.ctc_topo(4)
    G = cabi.Graph(g)
    for kw in ({"hash_ratio": 0.5}, {"max_active": 1}, {"min_active": -1},
               {"min_active": 30, "max_active": 30}):
        with pytest.raises(RuntimeError, match="Check failed"):
            cabi.Batch(G, **kw)  # faster-decoder.cc:24-28
    bt = cabi.Batch(G, max_batch=1)
    bt.batch = 1
    lp = synth.make_logprobs(3, 4, 0)
    with pytest.raises(RuntimeError, match="InitDecoding"):
        bt.advance_host([lp])  # faster-decoder.cc:128-129
    bt.init_decoding()
    bt.advance_host([lp])
    with pytest.raises(RuntimeError, match="num_frames_ready >= num_frames_decoded_"):
        bt.advance_host([lp[:2]])  # faster-decoder.cc:137
    # empty frontier: nothing survives -> GetBestPath false
    arcs = [(0, 1, 1, 0, 0.0)]
    g2 = synth.graph_from_arcs(2, 0, arcs, {})
    b2 = cabi.Batch(cabi.Graph(g2), max_batch=1)
    b2.init_decoding()
    b2.advance_host([np.zeros((3, 2), np.float32)])
    p = b2.best_paths()[0]
    assert not p.ok and not b2.reached_final()[0]


def test_simple_decoder_mode_matches_canonical_oracle(oracle):
    # SimpleDecoder semantics (plain beam, float-first cost sum, PruneToks) on the GPU
    cases = [(synth.ctc_topo(32), None, 32, 10.0)]
    for seed in range(3):
        cases.append((synth.random_graph(200 + 200 * seed, 3.0, 16, 0.25, 40 + seed), None, 16,
                      6.0 + 2 * seed))
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    cases.append((g, g["lexicon"], 40, 9.0))
    for g, lex, V, beam in cases:
        G, OG = cabi.Graph(g), oracle.Graph(g)
        B, T = 3, 45
        lps = [synth.make_logprobs(T - 5 * i, V, i, peak=5.0, lexicon=lex) for i in range(B)]
        sb = cabi.SimpleBatch(G, max_batch=B, beam=beam)
        sb.init_decoding()
        decs = [oracle.CanonSimpleDecoder(OG, beam=beam) for _ in range(B)]
        for d in decs:
            d.init_decoding()
        for t0 in range(0, T, 9):
            sb.advance_host(lps, max_num_frames=9)
            for i, d in enumerate(decs):
                d.advance_decoding(lps[i], max_num_frames=9)
                s, c = sb.frame_tokens(i, d.num_frames_decoded())
                so, co = d.tokens()
                assert s.tolist() == so.tolist() and c.tolist() == co.tolist(), (i, t0)
        frc = sb.final_relative_cost()
        rf = sb.reached_final()
        for use_final in (True, False):
            paths = sb.best_paths(use_final)
            for i, d in enumerate(decs):
                _assert_same_path(paths[i], d.get_best_path(use_final), (i, use_final))
                assert bool(rf[i]) == d.reached_final()
                assert frc[i] == np.float32(d.final_relative_cost()) or \
                    (np.isinf(frc[i]) and np.isinf(d.final_relative_cost()))
        # and against the reference-order restatement: same best words
        for i in range(B):
            r = oracle.SimpleDecoder(OG, beam=beam)
            r.decode(lps[i])
            assert r.get_best_path().osyms.tolist() == paths[i].osyms.tolist()


def test_candidate_queue_overflow_rescans_under_the_final_cutoff(oracle):
    # dense CTC topology: every frame scores 300 x 300 arcs, far more than the candidate queue
    # of a small spill table holds under the running cutoff; the frame is then scanned again
    # under the final cutoff, which few arcs survive
    g = synth.ctc_topo(300)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(8, 300, 0, peak=4.0)
    opts = dict(beam=3.0, min_active=0)
    bt = cabi.Batch(G, max_batch=1, table_slots=1 << 14, **opts)
    bt.init_decoding()
    bt.advance_host([lp])
    d = oracle.FasterDecoder(OG, policy=1, **opts)
    d.decode(lp)
    for f in (1, 4, 8):
        s, c = bt.frame_tokens(0, f)
        dd = oracle.FasterDecoder(OG, policy=1, **opts)
        dd.init_decoding()
        dd.advance_decoding(lp, max_num_frames=f)
        so, co = dd.tokens()
        assert s.tolist() == so.tolist() and c.tolist() == co.tolist(), f
    _assert_same_path(bt.best_paths()[0], d.get_best_path(), "rescan")


def test_one_call_decode_and_growing_pools_faster_decoder(oracle):
    """FasterDecoder::Decode as one launch over a work queue, from a token pool that is much
    too small to begin with; a vocabulary mismatch between graph and matrix is refused."""
    import torch
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, V = 7, 40
    lps = [synth.make_logprobs(30 + 9 * i, V, 50 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = [torch.from_numpy(lp).cuda() for lp in lps]
    bt = cabi.Batch(G, max_batch=B, beam=9.0, arena_tokens=2048, max_frames=4)
    bt.decode_device([d.data_ptr() for d in dev], [lp.shape[0] for lp in lps], V)
    assert bt.capacity()["grow_events"] > 0
    paths = bt.best_paths()
    for i in range(B):
        d = oracle.FasterDecoder(OG, beam=9.0, policy=1)
        d.decode(lps[i])
        _assert_same_path(paths[i], d.get_best_path(), i)
    with pytest.raises(RuntimeError, match="largest ilabel"):
        bt.decode_device([d.data_ptr() for d in dev], [lp.shape[0] for lp in lps], V - 1, row_stride=V)


def test_host_async_from_pinned_rows_that_are_refilled_at_once(oracle):
    """kd_batch_decode_host_async promises that the rows have left the caller's memory when it
    returns (ref: DecodableCtc copies its matrix, csrc/decodable-ctc.h:15-16).  From PINNED
    memory HIP's host-to-device copies are truly asynchronous, so the promise needs the
    library's own wait behind the last staging copy: a pipelined caller refills its buffer for
    the next call the moment this one returns.  Here the pinned rows are overwritten with
    garbage right after the call; the decode must still be that of the original rows."""
    import torch
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 16, 400, 40
    lps = [synth.make_logprobs(T, V, 90 + i, peak=6.0, lexicon=g["lexicon"]) for i in range(B)]
    pinned = torch.from_numpy(np.stack(lps)).pin_memory()
    assert pinned.is_pinned()
    views = [pinned[i].numpy() for i in range(B)]  # (no copy: the library reads the pinned pages)
    bt = cabi.Batch(G, max_batch=B, beam=8.0)
    want = []
    for lp in lps:
        d = oracle.FasterDecoder(OG, beam=8.0, policy=1)
        d.decode(lp)
        want.append(d.get_best_path())
    for _ in range(3):
        pinned.copy_(torch.from_numpy(np.stack(lps)))
        bt.decode_host(views, sync=False)
        pinned.fill_(-1e30)  # the caller's next batch, as far as the library may care
        bt.sync()
        paths = bt.best_paths()
        for i in range(B):
            _assert_same_path(paths[i], want[i], i)


def test_host_async_from_pageable_rows_that_are_refilled_at_once(oracle):
    """The same promise for PAGEABLE rows (plain numpy arrays), which the library does not
    wait for itself (ADVICE rounds 4-5): HIP's copy out of pageable memory has read the rows
    when it returns.  One matrix of the batch is pinned and sits where the library's few
    pointer look-ups do not necessarily probe first: a call that holds ANY pinned matrix among
    the probed ones must wait."""
    import torch
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 16, 400, 40
    lps = [synth.make_logprobs(T, V, 190 + i, peak=6.0, lexicon=g["lexicon"]) for i in range(B)]
    want = []
    for lp in lps:
        d = oracle.FasterDecoder(OG, beam=8.0, policy=1)
        d.decode(lp)
        want.append(d.get_best_path())
    bt = cabi.Batch(G, max_batch=B, beam=8.0)
    last_pinned = torch.from_numpy(lps[B - 1].copy()).pin_memory()
    for rep in range(3):
        rows = [lp.copy() for lp in lps]
        if rep == 2:
            last_pinned.copy_(torch.from_numpy(lps[B - 1]))
            rows[B - 1] = last_pinned.numpy()
        bt.decode_host(rows, sync=False)
        for r in rows:
            r.fill(-1e30)
        bt.sync()
        paths = bt.best_paths()
        for i in range(B):
            _assert_same_path(paths[i], want[i], i)


def test_best_path_longer_than_the_staging_queue_grows_the_queue(oracle):
    """GetBestPath stages the path in the workgroup's candidate queue (2 x table_slots entries).
    A path of 1000 + 40 x 1001 arcs -- a chain of a thousand epsilon arcs walked by InitDecoding
    and once more per frame --
    against a queue of 2 x 2^14: the utterance stops with KD_UTT_PATH_QUEUE_FULL (a status of
    its own: neither the token table nor the output pool is what ran out), the host grows the
    per-workgroup tables and asks again; with no_grow the message names the limit."""
    L, T = 1000, 40
    arcs = [(i, i + 1, 0, 0, 0.0) for i in range(L)] + [(L, 0, 1, 7, 0.5), (L, L, 2, 0, 3.0)]
    g = synth.graph_from_arcs(L + 1, 0, arcs, {L: 0.0})
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = np.zeros((T, 2), np.float32)
    bt = cabi.Batch(G, max_batch=1, beam=8.0, min_active=0, table_slots=1 << 14)
    slots0 = bt.capacity()["table_slots"]
    bt.decode_host([lp])
    p = bt.best_paths()[0]
    d = oracle.FasterDecoder(OG, beam=8.0, min_active=0, policy=1)
    d.decode(lp)
    _assert_same_path(p, d.get_best_path(), "long")
    assert len(p.ilabel) == L + T * (L + 1) > 2 * slots0
    assert bt.capacity()["table_slots"] > slots0
    fixed = cabi.Batch(G, max_batch=1, beam=8.0, min_active=0, table_slots=1 << 14, no_grow=1)
    with pytest.raises(RuntimeError, match="best path longer than the staging queue"):
        fixed.decode_host([lp])
        fixed.best_paths()


def test_largest_ilabel_the_arc_record_holds():
    """ilabel 2^20 - 1 (all twenty bits of the arc record's label field) next to ilabel 1, over a
    log-likelihood row of 2^20 - 1 columns (read in place: rows this wide are not staged in
    LDS): the label comes back whole, the right column was read."""
    big = (1 << 20) - 1
    arcs = [(0, 1, big, 5, 0.5), (0, 1, 1, 6, 0.25), (1, 1, big, 0, 0.0)]
    g = synth.graph_from_arcs(2, 0, arcs, {1: 0.0})
    G = cabi.Graph(g)
    lp = np.full((2, big), -8.0, np.float32)
    lp[0, big - 1] = -1.0   # frame 0: the big label is cheap (0.5 + 1 against 0.25 + 8)
    lp[1, big - 1] = -2.0
    bt = cabi.Batch(G, max_batch=1, beam=16.0, min_active=0)
    bt.decode_host([lp])
    p = bt.best_paths()[0]
    assert p.ok and p.ilabel.tolist() == [big, big] and p.olabel.tolist() == [5, 0]
    assert p.ac_cost.tolist() == [1.0, 2.0] and p.graph_cost.tolist() == [0.5, 0.0]
    lb = cabi.LatticeBatch(G, max_batch=1, beam=16.0, lattice_beam=10.0, min_active=0)
    lb.decode_host([lp])
    lat = lb.raw_lattices()[0]
    assert sorted(lat.ilabel.tolist()) == [1, big, big]
