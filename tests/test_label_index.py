"""The label index's layout and lookup (csrc/kd_label_index.h), on the host: the same
template the kernel instantiates over global memory, here over an array -- against
std::lower_bound and brute force (tests/native/label_index_check.cc)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_label_index_layout_and_lookup(tmp_path):
    exe = str(tmp_path / "label_index_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                    "-I", os.path.join(ROOT, "kaldi-decoder_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "tests", "native", "label_index_check.cc"),
                    "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
