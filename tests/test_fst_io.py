"""OpenFst binary reader / writer.  No real .fst file exists in
this image, so these are writer -> reader round trips plus checks of the header
bytes against OpenFst's published constants; reading files produced by OpenFst
itself is unverified."""
import struct

import numpy as np
import pytest

import kaldi_decoder as kd
from kaldi_decoder import synth


def _same(a, b):
    assert (a.start, a.num_states, a.num_arcs) == (b.start, b.num_states, b.num_arcs)
    for x, y in zip(a.arrays(), b.arrays()):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("fst_type", ["vector", "const"])
def test_round_trip(tmp_path, fst_type):
    for g in (synth.ctc_topo(7), synth.random_graph(300, 3.0, 20, 0.2, 1),
              synth.hl_graph(V=20, n_words=60, seed=3)):
        f = kd.fst_from_dict(g)
        path = str(tmp_path / f"g.{fst_type}.fst")
        f.write(path, fst_type)
        back = kd.Fst.read(path)
        _same(f, back)
        assert np.isinf(back.arrays()[5]).sum() == np.isinf(g["final"]).sum()


def test_header_layout(tmp_path):
    f = kd.fst_from_dict(synth.ctc_topo(3))
    path = str(tmp_path / "h.fst")
    f.write(path)
    raw = open(path, "rb").read()
    assert struct.unpack_from("<i", raw, 0)[0] == 2125659606          # kFstMagicNumber
    n = struct.unpack_from("<i", raw, 4)[0]
    assert raw[8:8 + n] == b"vector"
    off = 8 + n
    m = struct.unpack_from("<i", raw, off)[0]
    assert raw[off + 4:off + 4 + m] == b"standard"
    off += 4 + m
    version, flags = struct.unpack_from("<ii", raw, off)
    assert version == 2 and flags == 0
    start, ns, na = struct.unpack_from("<qqq", raw, off + 8 + 8)
    assert (start, ns, na) == (0, 3, 9)
    # first state: final weight (float) then int64 arc count
    body = off + 8 + 8 + 24
    assert struct.unpack_from("<f", raw, body)[0] == 0.0
    assert struct.unpack_from("<q", raw, body + 4)[0] == 3


def test_reader_rejects_garbage(tmp_path):
    p = tmp_path / "bad.fst"
    p.write_bytes(b"not an fst file at all")
    with pytest.raises(RuntimeError, match="bad magic"):
        kd.Fst.read(str(p))
    with pytest.raises(RuntimeError, match="cannot open"):
        kd.Fst.read(str(tmp_path / "missing.fst"))
    # a log-semiring file is refused
    f = kd.fst_from_dict(synth.ctc_topo(3))
    good = tmp_path / "g.fst"
    f.write(str(good))
    raw = bytearray(good.read_bytes())
    i = raw.find(b"standard")
    raw[i:i + 8] = b"log\0\0\0\0\0"
    bad = tmp_path / "log.fst"
    bad.write_bytes(bytes(raw))
    with pytest.raises(RuntimeError, match="arc type"):
        kd.Fst.read(str(bad))
