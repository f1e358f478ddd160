"""Pins the oracle's HashList restatement against the reference's own header.

oracle/_ref/libref_hashlist.so is the reference HashList<int32,int32>
(/root/reference/kaldi-decoder/csrc/hash-list.h, hash-list-inl.h) compiled in
place by oracle/Makefile.  The drive pattern follows the reference's own
randomised test (hash-list-test.cc:21-90): fill, then repeatedly Clear(),
SetSize() and re-Insert while walking the old list.
"""
import numpy as np
import pytest


def _drive(h, seed, rounds=40):
    rng = np.random.default_rng(seed)
    trace = []
    h.set_size(200)
    for _ in range(50):
        k, v = int(rng.integers(0, 200)), int(rng.integers(0, 50))
        trace.append(("ins", h.insert(k, v)))
    trace.append(("list", h.list()))
    for _ in range(rounds):
        old = h.clear()
        h.set_size(int(100 + rng.integers(0, 100)))
        for k, v in old:
            trace.append(("ins", h.insert(k + int(rng.integers(0, 3)), v)))
        for _ in range(10):
            trace.append(("find", h.find(int(rng.integers(0, 260)))))
        trace.append(("list", h.list()))
    trace.append(("final", h.clear()))
    return trace


@pytest.mark.parametrize("seed", range(6))
def test_iteration_order_matches_reference_header(oracle, seed):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref not built (needs /root/reference at build time)")
    assert _drive(oracle.HashList(ref=False), seed) == _drive(oracle.HashList(ref=True), seed)


def test_against_dict(oracle):
    # the reference test's own invariant: contents equal a plain map
    rng = np.random.default_rng(0)
    h = oracle.HashList()
    h.set_size(200)
    m = {}
    for _ in range(300):
        k, v = int(rng.integers(0, 500)), int(rng.integers(0, 1000))
        if h.insert(k, v):
            m[k] = v
        else:
            assert k in m
    assert dict(h.list()) == m
    for k in range(500):
        assert h.find(k) == m.get(k, -1)
    assert sorted(h.clear()) == sorted(m.items())
    assert h.list() == []
