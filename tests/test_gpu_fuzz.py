"""A short run of tools/fuzz_gpu.py: random graphs, options (beams, max / min_active,
prune_interval), batch sizes, pool sizes that must grow mid-launch, one-launch Decode and
chunked AdvanceDecoding, lattice and 1-best decoders, every result compared with the
oracle's canonical decoder.  (Minutes of it, with other seeds, are run by hand each round:
it found the periodic sweep overwriting the next frame's log-likelihood row on frames of
more than 6144 tokens, and a pool allocator that handed the same records to two
utterances when the pool ran out mid-launch.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_parity(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "12", str(seed)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", [21])
def test_randomised_parity_with_shrunk_capacities(seed):
    """The same runs against `libkd_hip_stress.so` (make stress; part of `make all`): the
    collector limited to 4 frames and 8 pieces per call -- so that collections are split
    over several calls and piece sizes are doubled until the pieces fit, which the shipped
    sizes only do beyond two million records -- and sweep windows of at most 3 frames with
    one link slot per lane."""
    lib = os.path.join(ROOT, "kaldi-decoder_amd", "python", "kaldi_decoder", "lib", "libkd_hip_stress.so")
    assert os.path.exists(lib), "make -C kaldi-decoder_amd stress"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "15", str(seed)],
                       capture_output=True, text=True, timeout=300,
                       env={**os.environ, "KD_HIP_LIB": "libkd_hip_stress.so"})
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize("seed", [31])
def test_randomised_parity_under_timing_perturbation(seed):
    """The same runs against `libkd_hip_jitter.so` (make jitter; part of `make all`): behind every
    workgroup barrier each wave sleeps a pseudo-random time of its own, up to thousands of
    cycles (kd_search.hip, KdJitter), with the stress build's shrunk tail capacities.  A phase
    that reads an LDS region which the next phase overwrites without a barrier in between
    -- round 3's SweepWindow race, found by reading after it had bitten one bench run in six --
    gives wrong lattices or faults here.  (Replaces a test that grepped the source for that
    one barrier.)"""
    lib = os.path.join(ROOT, "kaldi-decoder_amd", "python", "kaldi_decoder", "lib", "libkd_hip_jitter.so")
    assert os.path.exists(lib), "make -C kaldi-decoder_amd jitter"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "30", str(seed)],
                       capture_output=True, text=True, timeout=600,
                       env={**os.environ, "KD_HIP_LIB": "libkd_hip_jitter.so"})
    assert r.returncode == 0 and "fuzz ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_exports_are_byte_identical_under_timing_perturbation():
    """What a batch exports -- lattice arrays in their exported ORDER (finalized, at two prune
    intervals, and un-finalized from a stepwise decode), FinalRelativeCost, 1-best paths -- is
    the same bytes from the shipped library and from the timing-perturbation build, in which
    every wave passes every barrier at another time than in the shipped one and than in its own
    previous run: results, order included, do not depend on wave scheduling."""
    digests = []
    libs = ("libkd_hip.so", "libkd_hip_jitter_full.so", "libkd_hip_jitter.so")
    for lib in libs:  # (all three are part of `make all`: a missing one fails, it does not degrade)
        assert os.path.exists(os.path.join(ROOT, "kaldi-decoder_amd", "python", "kaldi_decoder", "lib", lib)), \
            f"{lib} is missing: make -C kaldi-decoder_amd all"
    for lib in libs:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "export_digest.py")],
                           capture_output=True, text=True, timeout=300, env={**os.environ, "KD_HIP_LIB": lib})
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("digest")]
        assert line, r.stdout
        digests.append(line[0])
    assert digests[0] == digests[1] == digests[2], digests
    assert int(digests[0].split()[-1]) > 1000
