"""The world > 1 branch of bench.py, executed: two ranks on the ONE GPU of the test box over
gloo (RCCL refuses two ranks on one device; gloo does not).  `bench.py --gpus 2` is started
plainly, as the driver's scaling step may start it, so bench.py launches its own
torch.distributed.run child; both ranks decode their own utterances, agree on the repeat
count (all_reduce), fence with barriers, push every call's lattices through ResultGatherer on
its own process group, the times are gathered, and --verify-gather has rank 0 decode rank 1's
utterances itself and compare what the gather delivered with that, array for array."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--workload", "small", "--backend", "gloo", "--one-device",
                        "--verify-gather", "--steps", "6", "--warmup", "1", "--steps-per-call", "2",
                        "--min-seconds", "0.5"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # (rank 0 prints, rank 1 does not)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gather_backend"] == "gloo"
    assert len(d["per_rank_value"]) == 2 and all(v > 0 for v in d["per_rank_value"])
    # whole-job value = both ranks' frames over the slower rank's time
    assert abs(d["value"] - 2 * min(d["per_rank_value"])) / d["value"] < 1e-6
    assert d["gather_ms_per_call"] is not None and d["gather_ms_per_call"] > 0
    v = d["gather_verified"]
    assert v == dict(ranks=2, utterances=2 * 2 * 16, mismatches=0)  # 2 ranks x 2 steps x B = 16
    # the CPU leg runs on rank 0 whatever the world size, and checks the GPU's 1-best
    assert d["cpu_baseline"]["cores"] == 1 and d["parity"]["one_best_mismatches"] == 0
    assert d["steps"] == 6 and d["steps_timed"] % 6 == 0


def _visible_gpus():
    import torch
    return torch.cuda.device_count()  # (counting devices does not initialise the GPU)


@pytest.mark.gpu
def test_bench_two_ranks_on_two_gpus_over_rccl():
    """Armed for the first box with two devices: the same run over
    the `nccl` backend -- RCCL -- one rank per GPU: `gather_exact`'s `send` of a device tensor
    against rank 0's grouped `irecv`, the worker thread's communicator beside the main one, and
    two graph uploads at once, inside GPUTEST rather than in a scaling run nobody can debug.
    On a one-GPU box it is skipped, with the reason."""
    n = _visible_gpus()
    if n < 2:
        pytest.skip("needs two GPUs for RCCL (RCCL refuses two ranks on one device); %d visible: "
                    "no multi-GPU measurement has been made" % n)
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--workload", "small", "--backend", "nccl",
                        "--verify-gather", "--steps", "6", "--warmup", "1", "--steps-per-call", "2",
                        "--min-seconds", "0.5"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["gather_backend"] == "nccl"
    assert len(d["per_rank_value"]) == 2 and all(v > 0 for v in d["per_rank_value"])
    assert d["gather_verified"] == dict(ranks=2, utterances=2 * 2 * 16, mismatches=0)
    assert d["parity"]["one_best_mismatches"] == 0
