"""The Python surface of the drop-in package, checked on CPU: same names,
keyword arguments, defaults and __str__ as the reference bindings
(/root/reference/kaldi-decoder/python/csrc/faster-decoder.cc:14-58,
decodable-itf.cc:46-52, decodable-ctc.cc:13-15)."""
import numpy as np
import pytest

import kaldi_decoder as kd
from kaldi_decoder import synth


def test_reference_names_are_exported():
    for name in ("DecodableCtc", "DecodableInterface", "FasterDecoder", "FasterDecoderOptions"):
        assert hasattr(kd, name)


def test_faster_decoder_options_defaults_attrs_and_str():
    o = kd.FasterDecoderOptions()
    assert (o.beam, o.max_active, o.min_active, o.beam_delta, o.hash_ratio) == \
        (16.0, 2**31 - 1, 20, 0.5, 2.0)
    # reference ToString(): faster-decoder.h:51-62
    assert str(o) == ("FasterDecoderOptions(beam=16, max_active=2147483647, min_active=20, "
                      "beam_delta=0.5, hash_ratio=2)")
    o = kd.FasterDecoderOptions(beam=12.5, max_active=3000, min_active=0, beam_delta=0.25,
                                hash_ratio=3.0)
    o.beam = 9.0
    o.max_active = 7
    assert str(o).startswith("FasterDecoderOptions(beam=9, max_active=7, min_active=0")


def test_decodable_ctc_semantics():
    lp = np.arange(12, dtype=np.float32).reshape(3, 4)
    d = kd.DecodableCtc(lp, offset=5)
    lp[:] = 0  # the constructor copied (decodable-ctc.h:15-16)
    assert d.num_frames_ready() == 8 and d.num_indices() == 4
    assert d.log_likelihood(6, 2) == 5.0  # row 1, column index-1
    assert d.is_last_frame(7) and not d.is_last_frame(6)
    assert isinstance(d, kd.DecodableInterface)


def test_decodable_interface_is_subclassable():
    class Mine(kd.DecodableInterface):
        def log_likelihood(self, frame, index):
            return -float(frame + index)

        def is_last_frame(self, frame):
            return frame == 2

        def num_frames_ready(self):
            return 3

        def num_indices(self):
            return 4

    m = Mine()
    assert m.log_likelihood(1, 2) == -3.0 and m.num_indices() == 4

    class NoReady(kd.DecodableInterface):
        def log_likelihood(self, frame, index):
            return 0.0

        def is_last_frame(self, frame):
            return True

        def num_indices(self):
            return 1

    with pytest.raises(RuntimeError, match="NumFramesReady"):
        NoReady().num_frames_ready()  # decodable-itf.h:90-94


def test_fst_and_lattice_types():
    g = synth.ctc_topo(4)
    f = kd.fst_from_dict(g)
    assert (f.start, f.num_states, f.num_arcs) == (0, 4, 16)
    assert f.final(1) == 0.0
    with pytest.raises(RuntimeError):
        kd.Fst(0, np.zeros(3, np.int32), [], [], [], [], [0.0])  # inconsistent sizes
    lat = kd.Lattice()
    assert lat.num_states == 0 and lat.start == -1
    ok, isyms, osyms, w = kd.get_linear_symbol_sequence(lat)
    assert ok and isyms == [] and osyms == []


def test_lattice_configs_defaults_and_str():
    # lattice-simple-decoder.h:37-48,68-83 ; python/csrc/lattice-simple-decoder.cc:15-32
    c = kd.LatticeSimpleDecoderConfig()
    assert (c.beam, c.lattice_beam, c.prune_interval, c.determinize_lattice, c.prune_lattice) == \
        (16.0, 10.0, 25, True, True)
    assert abs(c.beam_ratio - 0.9) < 1e-6 and abs(c.prune_scale - 0.1) < 1e-6
    assert str(c) == ("LatticeSimpleDecoderConfig(beam=16, lattice_beam=10, prune_interval=25, "
                      "determinize_lattice=True, prune_lattice=1, beam_ratio=0.9, prune_scale=0.1)")
    c.lattice_beam = 8.0
    assert "lattice_beam=8" in str(c)
    # lattice-faster-decoder.h:48-66,106-127
    f = kd.LatticeFasterDecoderConfig()
    assert (f.beam, f.max_active, f.min_active, f.lattice_beam, f.prune_interval) == \
        (16.0, 2**31 - 1, 200, 10.0, 25)
    assert (f.memory_pool_tokens_block_size, f.memory_pool_links_block_size) == (256, 256)
    assert str(f).startswith("LatticeFasterDecoderConfig(beam=16, max_active=2147483647, "
                             "min_active=200, lattice_beam=10, prune_interval=25, "
                             "determinize_lattice=True, beam_delta=0.5, hash_ratio=2, "
                             "prune_scale=0.1, memory_pool_tokens_block_size=256")
    for name in ("LatticeSimpleDecoder", "LatticeSimpleDecoderConfig", "LatticeFasterDecoder",
                 "LatticeFasterDecoderConfig"):
        assert hasattr(kd, name)


def test_shortest_path_on_a_hand_built_lattice():
    # no GPU involved: Lattice objects only come out of decoders, so go through
    # the empty case and the API shape here; the numeric check is in the GPU tests
    ok, best = kd.shortest_path(kd.Lattice())
    assert not ok and best.num_states == 0
