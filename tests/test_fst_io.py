"""OpenFst binary reader / writer.  No file written by OpenFst exists
in this image, so besides writer -> reader round trips the reader is given files built
here BYTE BY BYTE from OpenFst's published layout (fst.cc FstHeader::Write,
vector-fst.h / const-fst.h *Impl::Read, symbol-table.cc) -- independently of the C++
writer: with symbol tables attached, the aligned const container (file version 1 and the
IS_ALIGNED flag), the unaligned one, and a vector file whose header has no state count."""
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


# ---- files assembled from the published layout, not by the library's writer --------------
FST_MAGIC, SYMTAB_MAGIC = 2125659606, 2125658996
HAS_ISYMBOLS, HAS_OSYMBOLS, IS_ALIGNED = 1, 2, 4


def _s(text):  # OpenFst string: int32 length + bytes
    b = text.encode()
    return struct.pack("<i", len(b)) + b


def _symbol_table(name, symbols):
    out = struct.pack("<i", SYMTAB_MAGIC) + _s(name)
    out += struct.pack("<qq", len(symbols), len(symbols))  # available_key, size
    for key, sym in enumerate(symbols):
        out += _s(sym) + struct.pack("<q", key)
    return out


def _header(fst_type, version, flags, start, num_states, num_arcs, properties=0x0000956A5A950000):
    return (struct.pack("<i", FST_MAGIC) + _s(fst_type) + _s("standard") +
            struct.pack("<iiQqqq", version, flags, properties, start, num_states, num_arcs))


def _states_of(g):
    for st in range(g["num_states"]):
        a, b = int(g["row_ptr"][st]), int(g["row_ptr"][st + 1])
        yield st, [(int(g["ilabel"][k]), int(g["olabel"][k]), float(g["weight"][k]),
                    int(g["nextstate"][k])) for k in range(a, b)]


def _vector_file(g, flags=0, symtabs=b"", with_state_count=True):
    out = _header("vector", 2, flags, g["start"], g["num_states"] if with_state_count else -1,
                  len(g["ilabel"])) + symtabs
    for st, arcs in _states_of(g):
        out += struct.pack("<fq", float(g["final"][st]), len(arcs))
        for il, ol, w, ns in arcs:
            out += struct.pack("<iifi", il, ol, w, ns)
    return out


def _const_file(g, version, flags, symtabs=b""):
    aligned = bool(flags & IS_ALIGNED) or version == 1
    out = _header("const", version, flags, g["start"], g["num_states"], len(g["ilabel"])) + symtabs

    def pad(b):
        return b + b"\0" * ((16 - len(b) % 16) % 16) if aligned else b
    out = pad(out)
    for st, arcs in _states_of(g):
        out += struct.pack("<fIIII", float(g["final"][st]), int(g["row_ptr"][st]), len(arcs),
                           sum(a[0] == 0 for a in arcs), sum(a[1] == 0 for a in arcs))
    out = pad(out)
    for _, arcs in _states_of(g):
        for il, ol, w, ns in arcs:
            out += struct.pack("<iifi", il, ol, w, ns)
    return out


def _graphs():
    yield synth.ctc_topo(5)
    yield synth.random_graph(120, 3.0, 12, 0.25, 9)
    yield synth.hl_graph(V=16, n_words=40, seed=4)
    yield synth.graph_from_arcs(3, 1, [(1, 2, 1, 7, 0.5)], {2: 1.25})  # a state without arcs, start != 0


def _check(raw, g, tmp_path, name):
    p = tmp_path / name
    p.write_bytes(raw)
    f = kd.Fst.read(str(p))
    assert (f.start, f.num_states, f.num_arcs) == (g["start"], g["num_states"], len(g["ilabel"]))
    for got, key in zip(f.arrays(), ("row_ptr", "ilabel", "olabel", "weight", "nextstate", "final")):
        assert np.array_equal(got, np.asarray(g[key])), (name, key)
    return f


def test_reads_hand_assembled_vector_files(tmp_path):
    isyms = _symbol_table("tokens", ["<eps>", "<blk>", "a", "b"])
    osyms = _symbol_table("words", ["<eps>", "hello", "world"])
    for k, g in enumerate(_graphs()):
        _check(_vector_file(g), g, tmp_path, f"v{k}.fst")
        _check(_vector_file(g, HAS_ISYMBOLS | HAS_OSYMBOLS, isyms + osyms), g, tmp_path, f"vs{k}.fst")
        _check(_vector_file(g, HAS_OSYMBOLS, osyms), g, tmp_path, f"vo{k}.fst")
        _check(_vector_file(g, with_state_count=False), g, tmp_path, f"vn{k}.fst")


def test_reads_hand_assembled_const_files(tmp_path):
    isyms = _symbol_table("tokens", ["<eps>", "x"])
    for k, g in enumerate(_graphs()):
        _check(_const_file(g, 2, 0), g, tmp_path, f"c2{k}.fst")                 # unaligned
        _check(_const_file(g, 2, IS_ALIGNED), g, tmp_path, f"c2a{k}.fst")       # aligned by flag
        _check(_const_file(g, 1, 0), g, tmp_path, f"c1{k}.fst")                 # aligned by version
        _check(_const_file(g, 2, IS_ALIGNED | HAS_ISYMBOLS, isyms), g, tmp_path, f"c2s{k}.fst")


def test_writer_output_equals_the_hand_assembled_bytes(tmp_path):
    # ... and the other direction: what the library writes is what the layout says (the
    # properties word aside, which the writer leaves at "unknown")
    for k, g in enumerate(_graphs()):
        f = kd.fst_from_dict(g)
        for fst_type, want in (("vector", _vector_file(g)), ("const", _const_file(g, 2, IS_ALIGNED))):
            p = tmp_path / f"w{k}.{fst_type}"
            f.write(str(p), fst_type)
            got = bytearray(p.read_bytes())
            want = bytearray(want)
            off = 4 + 4 + len(fst_type) + 4 + len("standard") + 8  # magic, two strings, version, flags
            got[off:off + 8] = want[off:off + 8]
            assert bytes(got) == bytes(want), (k, fst_type)


def test_truncated_and_inconsistent_files_are_refused(tmp_path):
    g = synth.random_graph(50, 3.0, 8, 0.2, 2)
    raw = _vector_file(g)
    for cut in (10, len(raw) // 2, len(raw) - 3):
        p = tmp_path / f"t{cut}.fst"
        p.write_bytes(raw[:cut])
        with pytest.raises(RuntimeError, match="unexpected end of file"):
            kd.Fst.read(str(p))
    bad = dict(g)
    bad["nextstate"] = np.array(g["nextstate"])
    bad["nextstate"][3] = g["num_states"] + 5
    p = tmp_path / "ns.fst"
    p.write_bytes(_vector_file(bad))
    with pytest.raises(RuntimeError, match="nextstate out of range"):
        kd.Fst.read(str(p))


def test_decoder_constructors_take_graph_objects_and_paths(tmp_path):
    # (no GPU needed: construction fails later, at the device; the conversion is what is tested)
    g = synth.ctc_topo(4)

    class KaldifstLike:  # what kaldifst.StdVectorFst offers a caller: .write(path)
        def write(self, path):
            with open(path, "wb") as fh:
                fh.write(_const_file(g, 1, 0))

    f = kd.as_fst(KaldifstLike())
    assert (f.start, f.num_states, f.num_arcs) == (0, 4, 16)
    p = tmp_path / "g.fst"
    p.write_bytes(_vector_file(g))
    assert kd.as_fst(str(p)).num_arcs == 16 and kd.as_fst(p).num_arcs == 16
    assert kd.as_fst(f) is f
    with pytest.raises(TypeError, match="decoding graph"):
        kd.as_fst(42)
