"""GPU tests of the drop-in Python API, written the way icefall's CTC decode
scripts call the reference: DecodableCtc(log_probs), FasterDecoder(fst,
FasterDecoderOptions(max_active=3000)), decode, reached_final, get_best_path,
linear symbol sequence."""
import numpy as np
import pytest

import kaldi_decoder as kd
from kaldi_decoder import synth

pytestmark = pytest.mark.gpu


def _oracle_path(oracle, g, lp, **opts):
    d = oracle.FasterDecoder(oracle.Graph(g), policy=1, **opts)
    d.decode(lp)
    return d, d.get_best_path()


def test_icefall_style_decode_hl(oracle):
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    fst = kd.fst_from_dict(g)
    for utt in range(3):
        lp = synth.make_logprobs(60, 40, utt, peak=8.0, lexicon=g["lexicon"])
        decodable = kd.DecodableCtc(lp)
        decoder = kd.FasterDecoder(fst, kd.FasterDecoderOptions(max_active=3000))
        decoder.decode(decodable)
        assert decoder.num_frames_decoded() == 60
        ok, best_path = decoder.get_best_path()
        od, op = _oracle_path(oracle, g, lp, max_active=3000)
        assert ok and op.ok and decoder.reached_final() == od.reached_final()
        ok2, isyms, osyms, (gc, ac) = kd.get_linear_symbol_sequence(best_path)
        assert ok2
        assert isyms == op.isyms.tolist() and osyms == op.osyms.tolist()
        assert abs(gc - op.total_cost[0]) < 1e-4 * max(1, abs(gc))
        assert abs(ac - op.total_cost[1]) < 1e-4 * max(1, abs(ac))
        # RemoveEpsLocal left no eps:eps arcs on the linear path
        src, dst, il, ol, _, _ = best_path.arrays()
        assert not np.any((il == 0) & (ol == 0))
        assert best_path.num_states == len(il) + 1


def test_ctc_topo_token_ids(oracle):
    # H only: token ids are ilabel - 1, repeats collapse through the epsilon olabels
    V = 32
    g = synth.ctc_topo(V)
    lp = synth.make_logprobs(50, V, 4, peak=8.0)
    dec = kd.FasterDecoder(kd.fst_from_dict(g), kd.FasterDecoderOptions())
    dec.decode(kd.DecodableCtc(lp))
    ok, lat = dec.get_best_path()
    _, op = _oracle_path(oracle, g, lp)
    _, isyms, osyms, _ = lat.get_linear_symbol_sequence()
    assert ok and isyms == op.isyms.tolist() and osyms == op.osyms.tolist()
    assert len(isyms) == 50  # one emitting arc per frame


def test_python_subclassed_decodable_and_streaming(oracle):
    g = synth.hlg_graph(V=40, n_words=200, n_hist=200, avg_succ=8, seed=4)
    lp = synth.make_logprobs(45, 40, 2, peak=6.0, lexicon=g["lexicon"])

    class Mine(kd.DecodableInterface):
        def __init__(self, m):
            super().__init__()
            self.m = m
            self.calls = 0

        def log_likelihood(self, frame, index):
            self.calls += 1
            return float(self.m[frame, index - 1])

        def is_last_frame(self, frame):
            return frame == self.m.shape[0] - 1

        def num_frames_ready(self):
            return self.m.shape[0]

        def num_indices(self):
            return self.m.shape[1]

    fst = kd.fst_from_dict(g)
    opts = kd.FasterDecoderOptions(beam=10.0)
    a = kd.FasterDecoder(fst, opts)
    mine = Mine(lp)
    a.decode(mine)
    assert mine.calls == 45 * 40
    b = kd.FasterDecoder(fst, opts)
    b.init_decoding()
    while b.num_frames_decoded() < 45:
        b.advance_decoding(kd.DecodableCtc(lp), max_num_frames=7)
    c = kd.FasterDecoder(fst, opts)
    c.init_decoding()
    for s in range(0, 45, 10):
        c.advance_decoding(kd.DecodableCtc(lp[s:s + 10], offset=s))
    _, op = _oracle_path(oracle, g, lp, beam=10.0)
    for dec in (a, b, c):
        ok, lat = dec.get_best_path()
        _, isyms, osyms, _ = lat.get_linear_symbol_sequence()
        assert ok and isyms == op.isyms.tolist() and osyms == op.osyms.tolist()
    with pytest.raises(RuntimeError, match="InitDecoding"):
        kd.FasterDecoder(fst, opts).advance_decoding(kd.DecodableCtc(lp))
    assert kd.FasterDecoder(fst, opts).num_frames_decoded() == -1


def test_batch_decoder_and_device_tensors(oracle):
    import torch
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    fst = kd.fst_from_dict(g)
    B, T = 5, 40
    lps = [synth.make_logprobs(T - 3 * i, 40, i, peak=7.0, lexicon=g["lexicon"]) for i in range(B)]
    dev = [torch.from_numpy(lp).cuda() for lp in lps]
    bd = kd.BatchFasterDecoder(fst, kd.FasterDecoderOptions(beam=9.0), B)
    bd.decode([kd.decodable_from_tensor(t) for t in dev])
    assert bd.num_frames_decoded() == [lp.shape[0] for lp in lps]
    ok, lats = bd.get_best_path()
    rf = bd.reached_final()
    for i in range(B):
        od, op = _oracle_path(oracle, g, lps[i], beam=9.0)
        _, isyms, osyms, _ = lats[i].get_linear_symbol_sequence()
        assert ok[i] and rf[i] == od.reached_final()
        assert isyms == op.isyms.tolist() and osyms == op.osyms.tolist()
    # set_options takes effect on the next utterance
    bd.set_options(kd.FasterDecoderOptions(beam=4.0))
    bd.decode([kd.DecodableCtc(lp) for lp in lps])
    _, op = _oracle_path(oracle, g, lps[0], beam=4.0)
    assert bd.get_best_path()[1][0].get_linear_symbol_sequence()[2] == op.osyms.tolist()


def test_lattice_simple_decoder_python_flow(oracle):
    g = synth.hlg_graph(V=30, n_words=100, n_hist=100, avg_succ=6, seed=6)
    fst = kd.fst_from_dict(g)
    lp = synth.make_logprobs(50, 30, 1, peak=5.0, lexicon=g["lexicon"])
    config = kd.LatticeSimpleDecoderConfig(beam=9.0, lattice_beam=5.0)
    dec = kd.LatticeSimpleDecoder(fst, config)
    assert str(dec.get_config()) == str(config)
    reached = dec.decode(kd.DecodableCtc(lp))
    can = oracle.CanonLatticeDecoder(oracle.Graph(g), beam=9.0, lattice_beam=5.0)
    assert reached == can.decode(lp)
    assert dec.num_frames_decoded() == 50
    assert dec.final_relative_cost() == can.final_relative_cost()
    ok, raw = dec.get_raw_lattice()
    cl = can.get_raw_lattice()
    assert ok and raw.num_states == cl.num_states and raw.num_arcs == len(cl.src)
    src, dst, il, ol, gc, ac = raw.arrays()
    assert sorted(zip(il.tolist(), ol.tolist(), gc.tolist(), ac.tolist())) == \
        sorted(zip(cl.ilabel.tolist(), cl.olabel.tolist(), cl.graph_cost.tolist(), cl.ac_cost.tolist()))
    ok, best = dec.get_best_path()
    pc = can.get_best_path()
    ok2, isyms, osyms, (gcost, acost) = best.get_linear_symbol_sequence()
    assert ok and ok2 and isyms == pc.isyms.tolist() and osyms == pc.osyms.tolist()
    assert abs(gcost + acost - sum(pc.total_cost)) < 1e-3
    with pytest.raises(RuntimeError, match="use_final_probs == false"):
        dec.get_raw_lattice(False)
    # shortest_path of the raw lattice is what get_best_path returns
    ok3, sp = kd.shortest_path(raw)
    assert ok3 and sp.get_linear_symbol_sequence()[1:3] == (isyms, osyms)
    # the same best words as the 1-best decoder
    fd = kd.FasterDecoder(fst, kd.FasterDecoderOptions(beam=9.0, min_active=0))
    fd.decode(kd.DecodableCtc(lp))
    assert fd.get_best_path()[1].get_linear_symbol_sequence()[2] == osyms


def test_lattice_faster_decoder_streaming_and_batch(oracle):
    g = synth.hl_graph(V=30, n_words=200, seed=5)
    fst = kd.fst_from_dict(g)
    lps = [synth.make_logprobs(40, 30, i, peak=6.0, lexicon=g["lexicon"]) for i in range(3)]
    cfg = kd.LatticeFasterDecoderConfig(beam=10.0, lattice_beam=6.0, max_active=80, min_active=10)
    dec = kd.LatticeFasterDecoder(fst, cfg)
    dec.init_decoding()
    for s in range(0, 40, 8):
        dec.advance_decoding(kd.DecodableCtc(lps[0][s:s + 8], offset=s))
    assert dec.num_frames_decoded() == 40
    dec.finalize_decoding()
    ok, raw = dec.get_raw_lattice()
    can = oracle.CanonLatticeDecoder(oracle.Graph(g), beam=10.0, lattice_beam=6.0, max_active=80,
                                     min_active=10)
    can.decode(lps[0])
    cl = can.get_raw_lattice()
    assert ok and raw.num_states == cl.num_states and raw.num_arcs == len(cl.src)
    bd = kd.BatchLatticeDecoder(fst, cfg, 3)
    reached = bd.decode([kd.DecodableCtc(lp) for lp in lps])
    okb, lats = bd.get_raw_lattice()
    assert all(okb) and lats[0].num_arcs == raw.num_arcs and len(reached) == 3
    okp, paths = bd.get_best_path()
    for i in range(3):
        c = oracle.CanonLatticeDecoder(oracle.Graph(g), beam=10.0, lattice_beam=6.0, max_active=80,
                                       min_active=10)
        c.decode(lps[i])
        assert okp[i] and paths[i].get_linear_symbol_sequence()[2] == c.get_best_path().osyms.tolist()
    with pytest.raises(RuntimeError, match="Check failed"):
        kd.LatticeFasterDecoder(fst, kd.LatticeFasterDecoderConfig(prune_scale=2.0))
    with pytest.raises(RuntimeError, match="Check failed"):
        kd.LatticeSimpleDecoder(fst, kd.LatticeSimpleDecoderConfig(prune_interval=0))


def test_simple_decoder_python_flow(oracle):
    g = synth.hl_graph(V=40, n_words=400, seed=3)
    fst = kd.fst_from_dict(g)
    lp = synth.make_logprobs(50, 40, 2, peak=7.0, lexicon=g["lexicon"])
    dec = kd.SimpleDecoder(fst, 10.0)
    assert dec.decode(kd.DecodableCtc(lp)) is True
    ref = oracle.CanonSimpleDecoder(oracle.Graph(g), beam=10.0)
    ref.decode(lp)
    assert dec.num_frames_decoded() == 50 and dec.reached_final() == ref.reached_final()
    assert dec.final_relative_cost() == pytest.approx(ref.final_relative_cost())
    ok, lat = dec.get_best_path()
    _, isyms, osyms, _ = lat.get_linear_symbol_sequence()
    rp = ref.get_best_path()
    assert ok and isyms == rp.isyms.tolist() and osyms == rp.osyms.tolist()
    dec.init_decoding()
    dec.advance_decoding(kd.DecodableCtc(lp), max_num_frames=20)
    assert dec.num_frames_decoded() == 20
