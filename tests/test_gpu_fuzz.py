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
