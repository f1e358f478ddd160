"""kd_graph_create's epsilon-cycle check (csrc/kd_graph_check.h: in-degree peel, iterative
Tarjan, Karp's minimum cycle mean per component) against a brute-force statement of the same
rule on thousands of small random graphs -- compiled on its own with
-fsanitize=address,undefined, so that an index slip in the iterative SCC code shows as a
report, not as a wrong answer on some real graph later.

The rule (round 5): refused iff some epsilon cycle has a total weight of ZERO OR LESS -- a
negative arc on a cycle whose total stays positive is accepted (weight-pushed graphs carry
them).  Brute force: (min, +) powers of the epsilon arcs' weight matrix; a closed walk of at
most S arcs with total <= 0 exists iff a simple cycle with total <= 0 does."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rule(S, src, dst, il, w):
    eps = il == 0
    W = np.full((S, S), np.inf)
    for a in np.nonzero(eps)[0]:
        W[src[a], dst[a]] = min(W[src[a], dst[a]], float(w[a]))
    P = W.copy()
    for _ in range(S):
        if (np.diag(P) <= 0.0).any():
            return False
        P = (P[:, :, None] + W[None, :, :]).min(axis=1)
    return not (np.diag(P) <= 0.0).any()


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("gc") / "graph_check_driver")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-I", os.path.join(ROOT, "kaldi-decoder_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "graph_check_driver.cc"), "-o", exe], check=True)
    return exe


def test_epsilon_cycle_check_matches_the_rule_on_random_graphs(driver):
    rng = np.random.default_rng(7)
    blobs, want = [], []
    for case in range(3000):
        S = int(rng.integers(1, 14))
        A = int(rng.integers(0, 40))
        src = np.sort(rng.integers(0, S, A)).astype(np.int32)
        dst = rng.integers(0, S, A).astype(np.int32)
        il = np.where(rng.random(A) < 0.6, 0, rng.integers(1, 5, A)).astype(np.int32)
        # weights from a small set so that zero and negative arcs, and zero CYCLES, are common
        w = rng.choice(np.asarray([0.0, 0.0, 0.5, 1.0, -0.5], np.float32), A).astype(np.float32)
        if case % 3 == 0:
            w = np.abs(w)
        row_ptr = np.zeros(S + 1, np.int32)
        np.add.at(row_ptr, src + 1, 1)
        row_ptr = np.cumsum(row_ptr).astype(np.int32)
        blobs.append(np.asarray([S, A], np.int32).tobytes() + row_ptr.tobytes() + il.tobytes() +
                     dst.tobytes() + w.tobytes())
        want.append(_rule(S, src, dst, il, w))
    r = subprocess.run([driver], input=b"".join(blobs), capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    got = [ln.split(" ", 1) for ln in r.stdout.decode().splitlines()]
    assert len(got) == len(want)
    assert [g[0] == "1" for g in got] == want
    assert 0.2 < np.mean(want) < 0.8  # (both answers are common: the test has teeth)
    msgs = {g[1].split(" through")[0].split(" of")[0] for g in got if g[0] == "0"}
    assert len(msgs) >= 1


def test_epsilon_cycle_check_on_a_long_chain_and_a_big_cycle(driver):
    """Depth: a chain of 200 000 epsilon arcs (no recursion to overflow) closed into one cycle by
    a last arc -- positive (accepted), zero (refused), negative (refused)."""
    S = 200000
    row_ptr = np.arange(S + 1, dtype=np.int32)
    il = np.zeros(S, np.int32)
    dst = ((np.arange(S) + 1) % S).astype(np.int32)
    blobs = []
    for last in (0.25, 0.0, -1.0):
        w = np.zeros(S, np.float32)
        w[-1] = last
        blobs.append(np.asarray([S, S], np.int32).tobytes() + row_ptr.tobytes() + il.tobytes() +
                     dst.tobytes() + w.tobytes())
    r = subprocess.run([driver], input=b"".join(blobs), capture_output=True, timeout=120)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = r.stdout.decode().splitlines()
    assert out[0].startswith("1") and out[1].startswith("0") and "total weight zero" in out[1]
    assert out[2].startswith("0") and "below zero" in out[2]


def test_negative_arc_on_a_positive_cycle_is_accepted_and_nan_is_not(driver):
    """a weight-pushed graph may carry a negative epsilon arc inside a cycle
    whose total is positive -- accepted; the same cycle at total zero or below, or with a NaN
    weight, is refused."""
    def blob(ws):
        S = 3
        row_ptr = np.asarray([0, 1, 2, 3], np.int32)
        il = np.zeros(3, np.int32)
        dst = np.asarray([1, 2, 0], np.int32)
        return (np.asarray([S, 3], np.int32).tobytes() + row_ptr.tobytes() + il.tobytes() + dst.tobytes() +
                np.asarray(ws, np.float32).tobytes())
    cases = [[1.0, -0.75, 0.5], [1.0, -1.5, 0.5], [1.0, -1.75, 0.5], [1.0, float("nan"), 0.5]]
    r = subprocess.run([driver], input=b"".join(blob(c) for c in cases), capture_output=True, timeout=60)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    out = r.stdout.decode().splitlines()
    assert out[0].startswith("1")
    assert out[1].startswith("0") and "total weight zero" in out[1]
    assert out[2].startswith("0") and "below zero" in out[2]
    assert out[3].startswith("0") and "NaN" in out[3]


def test_ilabels_beyond_twenty_bits_are_not_refused():
    """Until round 6 kd_graph_create refused an ilabel that does not fit the arc record's 20 bits;
    now the field's largest value is an escape code and the label stands in a side array
    (kd_device.h kIlabelEscape; the decode: tests/test_kat.py).  Without a GPU the upload fails --
    not a label check.  A negative label is still refused (before the GPU is touched)."""
    from kaldi_decoder import cabi, synth
    for lab in ((1 << 20) - 1, 1 << 20, (1 << 31) - 1):
        g = synth.graph_from_arcs(2, 0, [(0, 1, lab, 0, 0.5)], {1: 0.0})
        try:
            cabi.Graph(g)
        except RuntimeError as e:
            assert "largest supported" not in str(e) and cabi.device_count() == 0, e
    g = synth.graph_from_arcs(2, 0, [(0, 1, 3, 0, 0.5)], {1: 0.0})
    g["ilabel"] = g["ilabel"].copy()
    g["ilabel"][0] = -1
    with pytest.raises(RuntimeError, match="negative ilabel"):
        cabi.Graph(g)
