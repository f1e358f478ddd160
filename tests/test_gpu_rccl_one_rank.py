"""RCCL is loaded and used at least once by this code: ONE rank on cuda:0 with backend "nccl"
(= RCCL on ROCm) runs the collectives bench.py issues on its device-resident control words
(`ctl_dev = "cuda"`): init_process_group with a device id, all_reduce(MAX) of the repeat count,
all_gather of the per-rank times and of a payload's length (sharded.gather_exact's first step),
barrier, a second process group for the gather thread, and a device payload through
sharded.ResultGatherer.  One GPU cannot hold two RCCL ranks (RCCL refuses: the two-rank run of
tests/test_gpu_bench_world2.py goes over gloo), so the point-to-point send / grouped receive of
gather_exact stays un-executed on RCCL until a multi-GPU node runs `bench.py --gpus N`."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import os, sys
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "kaldi-decoder_amd", "python")]
import torch
import torch.distributed as dist
from kaldi_decoder import sharded
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
ctl_dev = "cuda"
tr = torch.tensor([3], dtype=torch.int64, device=ctl_dev)
dist.all_reduce(tr, op=dist.ReduceOp.MAX)
assert int(tr.item()) == 3
t = torch.tensor([1.25], dtype=torch.float64, device=ctl_dev)
ts_all = [torch.zeros_like(t)]
dist.all_gather(ts_all, t)
assert float(ts_all[0].item()) == 1.25
payload = torch.arange(1000, dtype=torch.int64, device="cuda")
n = torch.tensor([payload.numel()], dtype=torch.int64, device=payload.device)
sizes = [torch.zeros_like(n)]
dist.all_gather(sizes, n)           # (gather_exact's size exchange, on device words)
assert int(torch.cat(sizes)[0]) == 1000
dist.barrier()
torch.cuda.synchronize()
grp = dist.new_group()              # (the gather thread's own communicator)
g = sharded.ResultGatherer(dst=0, group=grp, keep=True)
ev = torch.cuda.Event(); ev.record()
g.submit(payload, ev)
got = g.close()
assert len(got) == 1 and torch.equal(got[0][0], payload)
dist.barrier(group=grp)
dist.destroy_process_group()
print("rccl one rank ok", torch.cuda.nccl.version())
"""


@pytest.mark.gpu
def test_one_rank_rccl_collectives_on_device_control_words():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "rccl one rank ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    # the process mapped RCCL itself
    assert "rccl one rank ok" in r.stdout
