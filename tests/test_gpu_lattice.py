"""GPU parity of the lattice decoders against the oracle's canonical lattice
policy: identical lattices (as numbering-free arc sets with bit-equal float
costs), identical state sets, final weights and final_relative_cost."""
import numpy as np
import pytest

from kaldi_decoder import cabi, synth

pytestmark = pytest.mark.gpu


def _cases():
    for seed in range(3):
        yield (f"rand{seed}", synth.random_graph(150 + 200 * seed, 3.0, 16, 0.2, 300 + seed), None,
               16, dict(beam=6.0 + seed, lattice_beam=3.0 + seed, min_active=0))
    g = synth.hl_graph(V=30, n_words=200, seed=5)
    yield "HL", g, g["lexicon"], 30, dict(beam=10.0, lattice_beam=6.0, min_active=0)
    g = synth.hlg_graph(V=30, n_words=100, n_hist=100, avg_succ=6, seed=6)
    yield "HLG", g, g["lexicon"], 30, dict(beam=9.0, lattice_beam=5.0, min_active=0)
    yield "HLGfaster", g, g["lexicon"], 30, dict(beam=9.0, lattice_beam=5.0, max_active=60,
                                                 min_active=10)
    yield "H16", synth.ctc_topo(16), None, 16, dict(beam=8.0, lattice_beam=4.0, min_active=0)


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_raw_lattice_matches_canonical_oracle(oracle, case):
    name, g, lex, V, opts = case
    B, T = 3, 50
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lps = [synth.make_logprobs(T - 4 * i, V, i, peak=5.0, lexicon=lex) for i in range(B)]
    lb = cabi.LatticeBatch(G, max_batch=B, **opts)
    lb.init_decoding()
    lb.advance_host(lps)
    frc_before = lb.final_relative_cost()
    lb.finalize_decoding()
    lats = lb.raw_lattices()
    frc = lb.final_relative_cost()
    for i in range(B):
        oo = {k: v for k, v in opts.items() if k != "prune_interval"}
        can = oracle.CanonLatticeDecoder(OG, **oo)
        can.decode(lps[i])
        cl = can.get_raw_lattice()
        gl = lats[i]
        assert gl.ok == cl.ok, (name, i)
        assert gl.num_states == cl.num_states, (name, i)
        assert gl.state_set() == cl.state_set(), (name, i)
        assert gl.arc_set() == cl.arc_set(), (name, i)
        assert len(gl.src) == len(cl.src), (name, i)  # and no arc twice
        assert gl.final_set() == cl.final_set(), (name, i)
        assert gl.state_graph[gl.start] == g["start"] and gl.state_frame[gl.start] == 0
        assert frc[i] == can.final_relative_cost() == frc_before[i], (name, i)
        # states are frame-major
        assert np.all(np.diff(gl.state_frame) >= 0)


def test_lattice_superset_of_reference_order_decoder_minus_extras(oracle):
    # against the reference restatement itself (policy 0): GPU lattice is contained in it
    g = synth.hlg_graph(V=30, n_words=100, n_hist=100, avg_succ=6, seed=6)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(60, 30, 2, peak=5.0, lexicon=g["lexicon"])
    lb = cabi.LatticeBatch(G, max_batch=1, beam=9.0, lattice_beam=5.0, min_active=0)
    lb.init_decoding()
    lb.advance_host([lp])
    lb.finalize_decoding()
    gl = lb.raw_lattices()[0]
    ref = oracle.LatticeSimpleDecoder(OG, beam=9.0, lattice_beam=5.0)
    ref.decode(lp)
    rl = ref.get_raw_lattice()
    assert gl.arc_set() <= rl.arc_set()
    assert lb.final_relative_cost()[0] == ref.final_relative_cost()


def test_lattice_error_behaviour_and_unfinalized_export(oracle):
    g = synth.random_graph(100, 3.0, 8, 0.2, 5)
    G = cabi.Graph(g)
    with pytest.raises(RuntimeError, match="Check failed"):
        cabi.LatticeBatch(G, lattice_beam=0.0)  # lattice-faster-decoder.h:128-133
    with pytest.raises(RuntimeError, match="Check failed"):
        cabi.LatticeBatch(G, prune_scale=1.5)
    lb = cabi.LatticeBatch(G, max_batch=1, beam=8.0, lattice_beam=4.0, min_active=0)
    lp = synth.make_logprobs(20, 8, 0, peak=4.0)
    lb.init_decoding()
    with pytest.raises(RuntimeError, match="num_frames > 0"):
        lb.raw_lattices()  # lattice-simple-decoder.cc:610
    lb.advance_host([lp])
    raw_nf = lb.raw_lattices(use_final_probs=False)[0]   # allowed before FinalizeDecoding
    raw = lb.raw_lattices(use_final_probs=True)[0]
    assert raw.num_states == raw_nf.num_states and len(raw_nf.final_state) >= len(raw.final_state)
    lb.finalize_decoding()
    pruned = lb.raw_lattices()[0]
    assert pruned.arc_set() <= raw.arc_set() and pruned.num_states <= raw.num_states
    with pytest.raises(RuntimeError, match="use_final_probs == false"):
        lb.raw_lattices(use_final_probs=False)  # :594-597
    with pytest.raises(RuntimeError, match="decoding_finalized_"):
        lb.finalize_decoding()
    with pytest.raises(RuntimeError, match="FinalizeDecoding"):
        lb.advance_host([lp])
    lb.init_decoding()  # a new utterance resets everything
    lb.advance_host([lp])
    lb.finalize_decoding()
    assert lb.raw_lattices()[0].arc_set() == pruned.arc_set()


def test_lattice_frames_that_spill_to_the_global_table(oracle):
    # frontier far beyond the LDS table: the lattice path must survive the replay
    g = synth.random_graph(30000, 4.0, 16, 0.1, 9, max_weight=0.5)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(8, 16, 0, peak=0.0)
    opts = dict(beam=1000.0, lattice_beam=2.0, min_active=0)
    lb = cabi.LatticeBatch(G, max_batch=1, **opts)
    lb.init_decoding()
    lb.advance_host([lp])
    st = dict(zip(cabi.STAT_NAMES, lb.stats()[0].tolist()))
    assert st["spill_frames"] > 0 and st["small_frames"] > 0
    lb.finalize_decoding()
    gl = lb.raw_lattices()[0]
    can = oracle.CanonLatticeDecoder(OG, **opts)
    can.decode(lp)
    cl = can.get_raw_lattice()
    assert gl.state_set() == cl.state_set()
    assert gl.arc_set() == cl.arc_set()
    assert len(gl.src) == len(cl.src)  # no duplicate arcs


def _fan_graph(n, V):
    """Start state with `n` epsilon arcs (distinct weights) into states that each have two
    emitting arcs into a chain of final states: InitDecoding's epsilon closure alone holds
    n + 1 tokens."""
    arcs = []
    for i in range(n):
        arcs.append((0, 1 + i, 0, 0, 0.25 * (i % 7)))
        arcs.append((1 + i, n + 1 + (i % 5), 1 + i % V, 1 + i % 3, 0.5 * (i % 3)))
        arcs.append((1 + i, n + 1 + ((i + 2) % 5), 1 + (i + 1) % V, 0, 0.125))
    for j in range(5):
        arcs.append((n + 1 + j, n + 1 + (j + 1) % 5, 1 + j % V, 0, 0.25))
    return synth.graph_from_arcs(n + 6, 0, arcs, {n + 1 + j: 0.5 * j for j in range(5)})


@pytest.mark.parametrize("n", [3000, 7000])
def test_init_decoding_closure_beyond_the_lds_tables(oracle, n):
    # the start state's epsilon closure does not fit the LDS-resident tier (n = 3000: the
    # big LDS table) nor the LDS table at all (n = 7000: the global table, where a lattice
    # token's slot is claimed and given its cost by ONE compare-and-swap and its place in the
    # block is noted at insert -- InitDecoding takes that path through code of its own)
    V = 8
    g = _fan_graph(n, V)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lp = synth.make_logprobs(4, V, 3, peak=1.0)
    opts = dict(beam=1000.0, lattice_beam=1000.0, min_active=0)
    lb = cabi.LatticeBatch(G, max_batch=1, **opts)
    lb.decode_host([lp])
    can = oracle.CanonLatticeDecoder(OG, **opts)
    can.decode(lp)
    _assert_same_lattice(lb.raw_lattices()[0], can.get_raw_lattice(), f"fan {n}")
    assert len(lb.raw_lattices()[0].src) > n  # the closure's links are all there
    st = dict(zip(cabi.STAT_NAMES, lb.stats()[0].tolist()))
    assert (st["spill_frames"] > 0) == (n > 6144), st  # (the global table: InitDecoding included)
    # FasterDecoder through the same tiers (its slots keep the 64-bit cost key layout)
    bt = cabi.Batch(G, max_batch=1, beam=1000.0, min_active=0)
    bt.init_decoding()
    s0, c0 = bt.frame_tokens(0, 0)
    d = oracle.FasterDecoder(OG, policy=1, beam=1000.0, min_active=0)
    d.init_decoding()
    so, co = d.tokens()
    assert len(so) == n + 1 and s0.tolist() == so.tolist() and c0.tolist() == co.tolist()


def test_bench_size_workload_matches_canonical_oracle(oracle):
    # BASELINE configs[2] at full size (4.9 M states, beam 15, lattice beam 8, T = 250), a few
    # utterances: every tier of the search runs here (frontiers from 1 to > 20 000 tokens)
    import bench
    wl = bench.WORKLOADS["hlg-lattice"]
    g = synth.hlg_graph(**wl["graph"])
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 4, wl["frames"], wl["V"]
    lps = [synth.make_logprobs(T, V, i, peak=12.0, lexicon=g["lexicon"]) for i in range(B)]
    # (the bench turns periodic pruning off; here the reference's default interval, so that
    # the sweeps and the collector run at full size too)
    oopts = {k: v for k, v in wl["opts"].items() if k != "prune_interval"}
    lb = cabi.LatticeBatch(G, max_batch=B, max_frames=T, **{**oopts, "prune_interval": 25})
    lb.init_decoding()
    lb.advance_host(lps)
    st = lb.stats()
    assert st[:, 0].tolist() == [T] * B
    assert st[:, 11].sum() > 0 and st[:, 9].sum() > 0  # LDS-resident frames and spill frames
    lb.finalize_decoding()
    lats = lb.raw_lattices()
    frc = lb.final_relative_cost()
    for i in range(B):
        can = oracle.CanonLatticeDecoder(OG, **oopts)
        can.decode(lps[i])
        cl = can.get_raw_lattice()
        gl = lats[i]
        assert gl.state_set() == cl.state_set(), i
        assert gl.arc_set() == cl.arc_set() and len(gl.src) == len(cl.src), i
        assert gl.final_set() == cl.final_set(), i
        assert frc[i] == can.final_relative_cost(), i
        # size-independent properties: frame-major numbering, every arc goes forward in time
        # or stays in its frame, every state but the start has an incoming arc
        assert np.all(np.diff(gl.state_frame) >= 0)
        df = gl.state_frame[gl.dst] - gl.state_frame[gl.src]
        assert np.all((df == 0) | (df == 1))
        has_in = np.zeros(gl.num_states, bool)
        has_in[gl.dst] = True
        has_in[gl.start] = True
        assert has_in.all()


def _assert_same_lattice(gl, cl, tag):
    assert gl.ok == cl.ok, tag
    assert gl.num_states == cl.num_states, tag
    assert gl.state_set() == cl.state_set(), tag
    assert gl.arc_set() == cl.arc_set(), tag
    assert len(gl.src) == len(cl.src), tag
    assert gl.final_set() == cl.final_set(), tag
    assert np.all(np.diff(gl.state_frame) >= 0), tag  # frame-major numbering


@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_one_call_decode_matches_oracle_and_stepwise_path(oracle, case):
    """kd_batch_decode_async: InitDecoding + every frame + FinalizeDecoding + GetRawLattice in
    one launch (the workgroup that decoded an utterance prunes and exports it from the lists
    the backward sweep leaves) against the canonical oracle and against the call-by-call
    path; more utterances than the launch has workgroups would need > 256, so the work
    queue is exercised with ragged lengths and a batch object reused for a second call."""
    import torch
    name, g, lex, V, opts = case
    B, T = 5, 50
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lps = [synth.make_logprobs(T - 7 * i, V, 10 + i, peak=5.0, lexicon=lex) for i in range(B)]
    dev = [torch.from_numpy(lp).cuda() for lp in lps]
    lb = cabi.LatticeBatch(G, max_batch=B, **opts)
    for rep in range(2):
        order = list(range(B)) if rep == 0 else list(reversed(range(B)))
        lb.decode_device([dev[i].data_ptr() for i in order], [lps[i].shape[0] for i in order], V)
        assert lb.num_frames_decoded().tolist() == [lps[i].shape[0] for i in order]
        lats = lb.raw_lattices()
        frc = lb.final_relative_cost()
        step = cabi.LatticeBatch(G, max_batch=B, **opts)
        step.init_decoding(B)
        step.advance_host([lps[i] for i in order])
        step.finalize_decoding()
        slats = step.raw_lattices()
        for k, i in enumerate(order):
            can = oracle.CanonLatticeDecoder(OG, **opts)
            can.decode(lps[i])
            _assert_same_lattice(lats[k], can.get_raw_lattice(), (name, rep, i))
            _assert_same_lattice(lats[k], slats[k], (name, rep, i, "stepwise"))
            assert frc[k] == can.final_relative_cost()
        # after Decode() the reference refuses use_final_probs = false (:594-597)
        with pytest.raises(RuntimeError, match="FinalizeDecoding"):
            lb.raw_lattices(use_final_probs=False)
    # Decode() without the export, then GetRawLattice as a call of its own (general export)
    lb.decode_device([d.data_ptr() for d in dev], [lp.shape[0] for lp in lps], V, lattice=False)
    for i, lat in enumerate(lb.raw_lattices()):
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lps[i])
        _assert_same_lattice(lat, can.get_raw_lattice(), (name, "separate export", i))


def test_pools_and_tables_grow_on_demand(oracle):
    """Capacities are starting sizes: a token pool, link pool, output pool, spill table and
    frame index that are all far too small for the decode grow, the search resumes from the
    last committed frame, and the lattices are those of a batch that never ran short."""
    import torch
    g = synth.hlg_graph(V=40, n_words=300, n_hist=300, avg_succ=8, seed=12)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 6, 90, 40
    opts = dict(beam=11.0, lattice_beam=6.0, min_active=0)
    lps = [synth.make_logprobs(T, V, 30 + i, peak=4.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = [torch.from_numpy(lp).cuda() for lp in lps]
    small = cabi.LatticeBatch(G, max_batch=B, arena_tokens=4096, arena_links=4096, table_slots=1 << 14,
                              max_frames=16, **opts)
    cap0 = small.capacity()
    small.decode_device([d.data_ptr() for d in dev], [T] * B, V)
    cap1 = small.capacity()
    assert cap1["grow_events"] > 0 and cap1["arena_tokens"] > cap0["arena_tokens"]
    assert cap1["arena_links"] > cap0["arena_links"] and cap1["max_frames"] >= T
    lats = small.raw_lattices()
    for i in range(B):
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lps[i])
        _assert_same_lattice(lats[i], can.get_raw_lattice(), i)
    # streaming in chunks through the same small batch object (grows the frame index again)
    small2 = cabi.LatticeBatch(G, max_batch=B, arena_tokens=4096, arena_links=4096, max_frames=8, **opts)
    small2.init_decoding(B)
    for t0 in range(0, T, 13):
        small2.advance_host(lps, max_num_frames=13)
    small2.finalize_decoding()
    for i, lat in enumerate(small2.raw_lattices()):
        _assert_same_lattice(lat, lats[i], ("chunks", i))
    # no_grow restores the hard limits, naming the knob
    fixed = cabi.LatticeBatch(G, max_batch=B, arena_tokens=4096, arena_links=4096, no_grow=1, **opts)
    with pytest.raises(RuntimeError, match="pool full"):
        fixed.decode_device([d.data_ptr() for d in dev], [T] * B, V)


def test_pool_cursor_is_pulled_back_after_an_overshoot(oracle):
    """A take that does not fit pushes the pool's cursor past the capacity (one atomic add, never
    taken back: kd_search.hip PoolChunk), and every workgroup that runs into the end in the same
    launch pushes it further -- here by a frame's worth of tokens each (oversized requests: a
    frame of some ten thousand tokens against a pool of 1024).  The host pulls the cursor back
    to the capacity before it resumes (kd_api.hip ReadBack), so what the failed takes skipped is
    not lost: the pools end within a doubling of what the decode wrote, where they would
    otherwise have to outgrow the accumulated overshoot of a dozen relaunches."""
    import torch
    g = synth.random_graph(20000, 4.0, 16, 0.1, 77)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 8, 6, 16
    opts = dict(beam=20.0, lattice_beam=3.0, min_active=0, prune_interval=1 << 20)
    lps = [synth.make_logprobs(T, V, 60 + i, peak=0.5) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    lb = cabi.LatticeBatch(G, max_batch=B, arena_tokens=1024, arena_links=1024, **opts)
    lb.decode_tensor(dev)
    cap, st = lb.capacity(), lb.stats()
    wrote_tokens = int(st[:, 5].sum()) + B  # tokens_out of every frame + the start tokens
    wrote_links = int(st[:, 10].sum())
    assert cap["grow_events"] >= 8 and wrote_tokens > 100 * 1024
    assert cap["arena_tokens"] <= 2.5 * wrote_tokens, (cap, wrote_tokens)
    assert cap["arena_links"] <= 2.5 * wrote_links, (cap, wrote_links)
    lats = lb.raw_lattices()
    for i in (0, B - 1):
        can = oracle.CanonLatticeDecoder(OG, beam=20.0, lattice_beam=3.0)
        can.decode(lps[i])
        _assert_same_lattice(lats[i], can.get_raw_lattice(), i)


def test_more_utterances_than_workgroups_and_table_growth_mid_launch(oracle):
    """A launch has at most one workgroup per CU (256) and 300 utterances here: workgroups
    take second utterances from the work queue.  The frontier (tens of thousands of tokens
    on flat posteriors) overflows the 2^14-slot spill table the batch starts with, so
    utterances stop with a full table, their workgroups go on with others -- from a table
    that must be clean again -- and the host grows the tables and resumes the rest."""
    import torch
    g = synth.random_graph(40000, 4.0, 24, 0.1, 911)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 300, 7, 24
    opts = dict(beam=30.0, lattice_beam=6.0, min_active=0)
    rng = np.random.default_rng(5)
    base = [synth.make_logprobs(T, V, 200 + i, peak=0.5) for i in range(6)]
    pick = rng.integers(0, len(base), B)
    dev = [torch.from_numpy(lp).cuda() for lp in base]
    lb = cabi.LatticeBatch(G, max_batch=B, table_slots=1 << 14, **opts)
    lb.decode_device([dev[k].data_ptr() for k in pick], [T] * B, V)
    cap = lb.capacity()
    assert cap["workgroups"] <= 2 * 256 < 2 * B and cap["table_slots"] > (1 << 14) and cap["grow_events"] > 0
    assert lb.stats()[:, 8].max() > (1 << 14) * 3 // 4
    lats = lb.raw_lattices()
    want = []
    for lp in base:
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lp)
        want.append(can.get_raw_lattice())
    for i in range(B):
        cl, gl = want[pick[i]], lats[i]
        assert gl.num_states == cl.num_states and len(gl.src) == len(cl.src), i
        if i % 25 == 0:
            _assert_same_lattice(gl, cl, i)


@pytest.mark.parametrize("lazy", [0, 1], ids=["run", "owed"])
@pytest.mark.parametrize("prune_interval", [1, 2, 7])
@pytest.mark.parametrize("case", list(_cases()), ids=lambda c: c[0])
def test_periodic_pruning_and_collection_leave_the_lattice_unchanged(oracle, case, prune_interval, lazy):
    """PruneActiveTokens every prune_interval frames (lattice-simple-decoder.cc:57-59) and
    the collector that moves the survivors of older frames into the archive: whatever the
    interval, the finalized lattice is the canonical oracle's; so is the lattice exported
    without FinalizeDecoding when it is compared after the same pruning on the oracle's
    side -- i.e. as the set of arcs a later FinalizeDecoding keeps."""
    import torch
    name, g, lex, V, opts = case
    B, T = 4, 60
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lps = [synth.make_logprobs(T - 5 * i, V, 20 + i, peak=5.0, lexicon=lex) for i in range(B)]
    dev = [torch.from_numpy(lp).cuda() for lp in lps]
    want = []
    for lp in lps:
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lp)
        want.append((can.get_raw_lattice(), can.final_relative_cost()))
    # one launch
    # (lazy = 0: every prune when its frame comes, like the reference; 1: owed until it shows --
    # here at the un-finalized export in the middle of the chunked decode)
    lb = cabi.LatticeBatch(G, max_batch=B, prune_interval=prune_interval, **opts)
    lb.set_lazy_prune(lazy)
    lb.decode_device([d.data_ptr() for d in dev], [lp.shape[0] for lp in lps], V)
    for i, lat in enumerate(lb.raw_lattices()):
        _assert_same_lattice(lat, want[i][0], (name, prune_interval, i))
    assert lb.final_relative_cost().tolist() == [w[1] for w in want]
    # in chunks of 11 frames (sweeps and collections fall on and between launch boundaries),
    # un-finalized export in the middle, then FinalizeDecoding as a call of its own
    st = cabi.LatticeBatch(G, max_batch=B, prune_interval=prune_interval, **opts)
    st.set_lazy_prune(lazy)
    st.init_decoding(B)
    for t0 in range(0, T, 11):
        st.advance_host(lps, max_num_frames=11)
        if t0 == 22:
            mid = st.raw_lattices(use_final_probs=False)
            assert all(m.ok for m in mid)
    st.finalize_decoding()
    for i, lat in enumerate(st.raw_lattices()):
        _assert_same_lattice(lat, want[i][0], (name, prune_interval, i, "chunks"))


def test_periodic_pruning_bounds_memory(oracle):
    """A long utterance over a dense graph in pools that may not grow: with the reference's
    default prune_interval the collector keeps the footprint at the pruned lattice plus a
    window of recent frames and the decode finishes; with pruning off the same pools are
    exhausted after a fraction of the frames."""
    import torch
    g = synth.ctc_topo(100)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    T, V = 1500, 100
    lp = synth.make_logprobs(T, V, 77, peak=7.0)
    dev = torch.from_numpy(lp).cuda()
    opts = dict(beam=8.0, lattice_beam=4.0, min_active=0)
    kw = dict(arena_tokens=1 << 16, arena_links=1 << 18, max_frames=T, no_grow=1)
    off = cabi.LatticeBatch(G, max_batch=1, prune_interval=10**6, **kw, **opts)
    with pytest.raises(RuntimeError, match="pool full"):
        off.decode_device([dev.data_ptr()], [T], V)
    on = cabi.LatticeBatch(G, max_batch=1, prune_interval=25, **kw, **opts)
    on.set_lazy_prune(0)  # (every prune when its frame comes, like the reference: tests/test_gpu_lazy_prune.py has the other way)
    on.decode_device([dev.data_ptr()], [T], V)
    assert on.num_frames_decoded().tolist() == [T]
    can = oracle.CanonLatticeDecoder(OG, **opts)
    can.decode(lp)
    _assert_same_lattice(on.raw_lattices()[0], can.get_raw_lattice(), "bounded")
    created = can.links_created()
    cap = on.capacity()
    assert cap["links_used"] < created // 2, (cap, created)  # the pool never held them all


def test_streaming_in_chunks_of_prune_interval_still_prunes(oracle):
    """AdvanceDecoding in chunks that end on multiples of prune_interval (25 frames a call):
    the prune a call leaves pending is performed by the next one, before its first frame,
    as the reference's Decode() does at the top of every frame
    (lattice-simple-decoder.cc:57-59) -- the footprint stays bounded and the finalized
    lattice is the canonical one."""
    import torch
    g = synth.ctc_topo(100)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    T, V = 1500, 100
    lp = synth.make_logprobs(T, V, 77, peak=7.0)
    dev = torch.from_numpy(lp).cuda()
    opts = dict(beam=8.0, lattice_beam=4.0, min_active=0)
    kw = dict(arena_tokens=1 << 16, arena_links=1 << 18, max_frames=T, no_grow=1)
    for chunk in (25, 5, 50):
        lb = cabi.LatticeBatch(G, max_batch=1, prune_interval=25, **kw, **opts)
        lb.set_lazy_prune(0)  # (the footprint bound asserted below is the eager prunes')
        lb.init_decoding()
        while lb.num_frames_decoded()[0] < T:
            lb.advance_device([dev.data_ptr()], [T], V, max_num_frames=chunk)
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lp)
        cap = lb.capacity()
        assert cap["links_used"] < can.links_created() // 2, (chunk, cap)
        lb.finalize_decoding()
        _assert_same_lattice(lb.raw_lattices()[0], can.get_raw_lattice(), f"chunk {chunk}")


def test_reached_final_on_a_lattice_batch_leaves_the_export_alone(oracle):
    """kd_batch_reached_final on a lattice batch is FinalRelativeCost() != infinity
    (lattice-simple-decoder.h:142-147) and must not disturb an exported lattice."""
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    lps = [synth.make_logprobs(40, 40, i, peak=6.0, lexicon=g["lexicon"]) for i in range(3)]
    # (the last utterance: three frames only -- no final state is active yet)
    lps[2] = lps[2][:3]
    opts = dict(beam=10.0, lattice_beam=6.0, min_active=0)
    lb = cabi.LatticeBatch(G, max_batch=3, **opts)
    lb.decode_host(lps)
    rf = lb.reached_final()
    lats = lb.raw_lattices()
    rf2 = lb.reached_final()
    for i, lp in enumerate(lps):
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lp)
        want = can.final_relative_cost() != float("inf")
        assert bool(rf[i]) == want and bool(rf2[i]) == want, i
        _assert_same_lattice(lats[i], can.get_raw_lattice(), f"utt {i}")
    with pytest.raises(RuntimeError, match="back-pointers"):
        lb.best_paths()


def test_decode_tensor_padded_batch_with_lengths(oracle):
    """The [B, T, V] (+ lengths) entry: one padded device tensor, ragged
    lengths, a non-contiguous view (every second time step of a longer tensor) -- same
    lattices as the canonical oracle on each utterance's valid rows; tolerance 0."""
    import torch
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 5, 48, 40
    lens = [48, 31, 1, 17, 40]
    lps = [synth.make_logprobs(T, V, i, peak=6.0, lexicon=g["lexicon"]) for i in range(B)]
    opts = dict(beam=10.0, lattice_beam=6.0, min_active=0)
    dense = torch.from_numpy(np.stack(lps)).cuda()
    wide = torch.zeros(B, 2 * T, V + 24, device="cuda")
    wide[:, ::2, :V] = dense
    for t in (dense, wide[:, ::2, :V]):
        lb = cabi.LatticeBatch(G, max_batch=B, **opts)
        lb.decode_tensor(t, torch.tensor(lens))
        assert lb.num_frames_decoded().tolist() == lens
        lats = lb.raw_lattices()
        for i in range(B):
            can = oracle.CanonLatticeDecoder(OG, **opts)
            can.decode(lps[i][:lens[i]])
            _assert_same_lattice(lats[i], can.get_raw_lattice(), f"utt {i}")
    bt = cabi.Batch(G, max_batch=B, beam=10.0)
    bt.decode_tensor(dense)  # lengths default to T
    paths = bt.best_paths()
    for i in range(B):
        d = oracle.FasterDecoder(OG, beam=10.0, policy=1)
        d.decode(lps[i])
        assert paths[i].ilabel.tolist() == d.get_best_path().ilabel.tolist()
    with pytest.raises(ValueError, match="unit stride"):
        lb.decode_tensor(wide[:, :T, ::2])


_RAW_FIELDS = ("src", "dst", "ilabel", "olabel", "graph_cost", "ac_cost", "state_frame",
               "state_graph", "final_state", "final_graph")


def _same_arrays(a, b):
    return (a.ok == b.ok and a.num_states == b.num_states and a.start == b.start and
            all(getattr(a, k).tobytes() == getattr(b, k).tobytes() for k in _RAW_FIELDS))


def test_exported_lattices_are_the_same_arrays_run_to_run(oracle):
    """GetRawLattice is deterministic in the reference (token-list and link-list order,
    lattice-simple-decoder.cc:584-657): same input, same arrays.  Here the order in which
    the search and the sweeps come across tokens and links depends on wave scheduling, so
    the export sorts: states by (frame, graph state), arcs by (source state, graph arc),
    finals by state.  The same batch decoded in one launch, again in one launch, stepwise
    (Init / Advance in chunks / Finalize / GetRawLattice: the general export, ordered on
    the host) and with another prune_interval must give byte-identical arrays; so must the
    un-finalized export twice."""
    import torch
    g = synth.hlg_graph(V=60, n_words=400, n_hist=400, avg_succ=10, seed=7)
    G = cabi.Graph(g)
    B, T, V = 6, 80, 60
    lps = [synth.make_logprobs(T, V, 20 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    opts = dict(beam=11.0, lattice_beam=7.0, min_active=0)

    def one_launch(**kw):
        lb = cabi.LatticeBatch(G, max_batch=B, **{**opts, **kw})
        lb.decode_tensor(dev)
        return [lb.raw_lattices()[i] for i in range(B)], lb

    def stepwise(chunk, **kw):
        lb = cabi.LatticeBatch(G, max_batch=B, **{**opts, **kw})
        lb.init_decoding()
        while lb.num_frames_decoded()[0] < T:
            lb.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, V, max_num_frames=chunk)
        raw = [lb.raw_lattices()[i] for i in range(B)]       # un-finalized: the general export
        raw2 = [lb.raw_lattices(False)[i] for i in range(B)]  # (forces another export launch)
        raw3 = [lb.raw_lattices()[i] for i in range(B)]
        lb.finalize_decoding()
        return [lb.raw_lattices()[i] for i in range(B)], raw, raw3

    a, lb_a = one_launch()
    b, _ = one_launch()
    c, _ = one_launch(prune_interval=7)
    d, d_raw, d_raw3 = stepwise(13)
    e, e_raw, _ = stepwise(80, prune_interval=10**6)
    for i in range(B):
        assert a[i].ok and len(a[i].src) > 20
        assert _same_arrays(a[i], b[i]), i
        assert _same_arrays(a[i], c[i]), i
        assert _same_arrays(a[i], d[i]), i
        assert _same_arrays(a[i], e[i]), i
        assert _same_arrays(d_raw[i], d_raw3[i]), i
        # the canonical order itself
        f, sg = a[i].state_frame.astype(np.int64), a[i].state_graph.astype(np.int64)
        key = f << 32 | sg
        assert (np.diff(key) > 0).all()
        assert (np.diff(a[i].src.astype(np.int64)) >= 0).all()
        assert (np.diff(a[i].final_state.astype(np.int64)) > 0).all()


def test_lattices_exported_into_a_lent_device_buffer(oracle):
    """kd_batch_set_output_buffer: the kernel writes the exported lattices into a device
    tensor of the caller's (what the multi-GPU gather hands to RCCL); buffer + table unpack to
    the very arrays raw_lattices() gives -- of this batch (through the host mirror) and of a
    batch that exports into its own pinned pool.  A buffer that is too small is an error."""
    import torch
    from kaldi_decoder import sharded
    g = synth.hlg_graph(V=60, n_words=400, n_hist=400, avg_succ=10, seed=7)
    G = cabi.Graph(g)
    B, T, V = 6, 80, 60
    lps = [synth.make_logprobs(T, V, 20 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = torch.from_numpy(np.stack(lps)).cuda()
    opts = dict(beam=11.0, lattice_beam=7.0, min_active=0)
    ref = cabi.LatticeBatch(G, max_batch=B, **opts)
    ref.decode_tensor(dev)
    want = ref.raw_lattices()
    lb = cabi.LatticeBatch(G, max_batch=B, **opts)
    buf = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    lb.set_output_buffer(buf)
    for _ in range(2):  # (a second call reuses the buffer from its start)
        lb.decode_tensor(dev)
        used, table = lb.raw_lattice_table()
        assert 0 < used <= buf.numel() and table.shape == (B, 5)
        got = sharded.unpack_pool(sharded.pool_payload(buf, used, table))
        mine = lb.raw_lattices()
        for i in range(B):
            assert _same_arrays(got[i], want[i]), i
            assert _same_arrays(mine[i], want[i]), i
    # an export that runs APART from the decode (step-wise API: advance, finalize, then
    # GetRawLattice) goes through the general, arena-ordered export, which the host puts in
    # order: raw_lattice_table mirrors the ordered regions into the lent buffer
    buf.zero_()
    lb.init_decoding(B)
    lb.advance_device([dev[i].data_ptr() for i in range(B)], [T] * B, V)
    lb.finalize_decoding()
    mine = lb.raw_lattices()
    used, table = lb.raw_lattice_table()
    got = sharded.unpack_pool(sharded.pool_payload(buf, used, table))
    for i in range(B):
        assert _same_arrays(mine[i], want[i]), i
        assert _same_arrays(got[i], want[i]), i
    lb.set_output_buffer(torch.zeros(64, dtype=torch.int64, device="cuda"))
    with pytest.raises(RuntimeError, match="output buffer"):
        lb.decode_tensor(dev)
    lb.set_output_buffer(None)
    lb.decode_tensor(dev)
    assert _same_arrays(lb.raw_lattices()[0], want[0])


def test_fused_log_softmax_matches_decoding_the_log_probs(oracle):
    """kd_batch_set_log_softmax: the batch is handed LOGITS and normalises each
    row where it stages it in LDS.  Against a decode of torch.log_softmax(logits): the same
    1-best labels and states, per-arc costs and FinalRelativeCost within 1e-4 relative (float
    rounding of another summation order; the tolerance north_star states for costs), for the
    lattice decoder and the 1-best decoder, one launch and stepwise.  Switched off again the
    batch is bit-exact with the oracle."""
    import torch
    g = synth.hlg_graph(V=60, n_words=400, n_hist=400, avg_succ=10, seed=7)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 5, 70, 60
    rng = np.random.default_rng(5)
    lps = [synth.make_logprobs(T, V, 50 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(B)]
    # logits: the log-probs shifted by an arbitrary amount per row and scaled back by softmax
    logits = torch.from_numpy(np.stack(lps)) + torch.from_numpy(rng.normal(0, 3, (B, T, 1)).astype(np.float32))
    logits = logits.cuda()
    ref_lp = torch.log_softmax(logits, dim=-1)

    def close(a, b):
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        if a.shape != b.shape or not np.array_equal(np.isinf(a), np.isinf(b)):
            return False
        fin = np.isfinite(b)
        return bool(np.all(np.abs(a[fin] - b[fin]) <= 1e-4 * np.maximum(1.0, np.abs(b[fin]))))

    bt_ref, bt = cabi.Batch(G, max_batch=B, beam=11.0), cabi.Batch(G, max_batch=B, beam=11.0)
    bt_ref.decode_tensor(ref_lp)
    bt.set_log_softmax(True)
    bt.decode_tensor(logits)
    for a, b in zip(bt.best_paths(), bt_ref.best_paths()):
        assert a.ilabel.tolist() == b.ilabel.tolist() and a.state.tolist() == b.state.tolist()
        assert close(a.ac_cost, b.ac_cost) and close(a.graph_cost, b.graph_cost)
    opts = dict(beam=11.0, lattice_beam=7.0, min_active=0)
    lb_ref, lb = cabi.LatticeBatch(G, max_batch=B, **opts), cabi.LatticeBatch(G, max_batch=B, **opts)
    lb_ref.decode_tensor(ref_lp)
    lb.set_log_softmax(True)
    lb.init_decoding(B)
    while lb.num_frames_decoded()[0] < T:  # stepwise: every staging site of the row
        lb.advance_device([logits[i].data_ptr() for i in range(B)], [T] * B, V, max_num_frames=9)
    lb.finalize_decoding()
    assert close(lb.final_relative_cost(), lb_ref.final_relative_cost())
    for i in range(B):
        a, b = lb.raw_lattices()[i], lb_ref.raw_lattices()[i]
        # (a link a hair's breadth from the lattice beam may fall on either side of it)
        common = {x[:6] for x in a.arc_set()} & {x[:6] for x in b.arc_set()}
        assert len(common) >= 0.98 * max(len(a.src), len(b.src)), i
    lb.set_log_softmax(False)
    lb.decode_tensor(ref_lp)
    for i in range(B):
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(ref_lp[i].cpu().numpy())
        _assert_same_lattice(lb.raw_lattices()[i], can.get_raw_lattice(), f"off {i}")


def test_decode_from_host_matrices_in_one_launch(oracle):
    """kd_batch_decode with KD_MEM_HOST: the matrices are staged, then the same one-launch
    Decode (ragged lengths); lattice and 1-best batches."""
    g = synth.hlg_graph(V=30, n_words=100, n_hist=100, avg_succ=6, seed=6)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    opts = dict(beam=9.0, lattice_beam=5.0, min_active=0)
    lps = [synth.make_logprobs(55 - 6 * i, 30, 60 + i, peak=5.0, lexicon=g["lexicon"]) for i in range(4)]
    lb = cabi.LatticeBatch(G, max_batch=4, **opts)
    lb.decode_host(lps)
    for i, lat in enumerate(lb.raw_lattices()):
        can = oracle.CanonLatticeDecoder(OG, **opts)
        can.decode(lps[i])
        _assert_same_lattice(lat, can.get_raw_lattice(), i)
    bt = cabi.Batch(G, max_batch=4, beam=9.0)
    bt.decode_host(lps)
    paths = bt.best_paths()
    for i in range(4):
        d = oracle.FasterDecoder(OG, beam=9.0, policy=1)
        d.decode(lps[i])
        p = d.get_best_path()
        assert paths[i].ilabel.tolist() == p.ilabel.tolist() and paths[i].ac_cost.tolist() == p.ac_cost.tolist()


def test_two_host_threads_decode_on_batches_of_their_own_over_one_graph(oracle):
    """The reference's contract (threading): a decoder instance is not re-entrant,
    distinct instances may share one FST read-only.  Here: one kd_graph, two host threads, each
    with batch objects and streams of its own (ctypes releases the GIL around the C calls, so
    the two threads are in the library at the same time: enqueueing, growing pools that start
    too small, reading results), a dozen calls each; an error raised in one thread must not
    show up in the other (kd_last_error is thread-local)."""
    import threading
    import torch
    g = synth.hlg_graph(V=60, n_words=400, n_hist=400, avg_succ=10, seed=7)
    G, OG = cabi.Graph(g), oracle.Graph(g)
    B, T, V = 8, 60, 60
    opts = dict(beam=11.0, lattice_beam=6.0, min_active=0)
    sets = [[synth.make_logprobs(T, V, 900 + 50 * k + i, peak=4.0, lexicon=g["lexicon"]) for i in range(B)]
            for k in range(2)]
    want = []
    for lps in sets:
        w = []
        for lp in lps:
            can = oracle.CanonLatticeDecoder(OG, beam=11.0, lattice_beam=6.0)
            can.decode(lp)
            w.append(can.get_raw_lattice())
        want.append(w)
    errors, bad = [], []

    def work(k):
        try:
            st = torch.cuda.Stream()
            dev = torch.from_numpy(np.stack(sets[k])).cuda()
            lb = cabi.LatticeBatch(G, max_batch=B, stream=st.cuda_stream, arena_tokens=2048, arena_links=2048,
                                   **opts)
            for it in range(12):
                if it % 4 == 3:  # an error of this thread's own, in between
                    try:
                        lb.decode_device([dev[i].data_ptr() for i in range(B)], [T] * B, 3)
                        bad.append((k, it, "no error for a matrix of 3 columns"))
                    except RuntimeError as e:
                        if "columns" not in str(e):
                            bad.append((k, it, str(e)))
                    continue
                lb.decode_tensor(dev)
                lats = lb.raw_lattices()
                for i in range(B):
                    if lats[i].arc_set() != want[k][i].arc_set() or lats[i].state_set() != want[k][i].state_set():
                        bad.append((k, it, i))
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors and not bad, (errors, bad[:5])
