"""CPU-side checks of the C ABI: the library loads, exports every symbol that
include/kd_hip.h declares, and validates arguments without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from kaldi_decoder import cabi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "kd_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kd_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    L = cabi.lib()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(cabi.SYMBOLS) == names


def test_default_options_match_reference_defaults():
    # /root/reference/kaldi-decoder/csrc/faster-decoder.h:41-49
    o = cabi.default_options()
    assert (o.beam, o.max_active, o.min_active, o.beam_delta, o.hash_ratio) == \
        (16.0, 2**31 - 1, 20, 0.5, 2.0)


def test_argument_validation_without_gpu():
    L = cabi.lib()
    out = C.c_void_p()
    assert L.kd_batch_create(None, None, 1, None, C.byref(out)) != 0
    assert b"bad argument" in L.kd_last_error()
    assert L.kd_batch_sync(None) != 0


def test_no_cpu_fallback_in_product_sources():
    # the product tree must not reference the oracle
    for base, _, files in os.walk(os.path.join(ROOT, "kaldi-decoder_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cc", ".cpp")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                assert "kd_oracle" not in txt and "import oracle" not in txt \
                    and "from oracle" not in txt, os.path.join(base, f)
