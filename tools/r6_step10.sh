#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step11
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lazy_prune.py tests/test_gpu_lattice.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -6
timeout 300 python tools/min200_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -4
timeout 120 python tools/fuzz_gpu.py 45 10201 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_stress.so timeout 120 python tools/fuzz_gpu.py 45 10202 2>&1 | tail -1
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python - $O/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f M  eager %.2f  off %.2f  min200 %.2f  large %s  single %.2f ms  launch %.1f alone %.1f" % (
    d["value"] / 1e6, (d.get("value_prune_eager") or 0) / 1e6, (d.get("value_prune_interval_off") or 0) / 1e6,
    (d.get("value_min_active_200") or 0) / 1e6, (d.get("large_lm") or {}).get("value"), d["single_batch_ms"],
    d["roofline"]["kernel_ms"], d["roofline"]["launch_alone"]["kernel_ms"]))
print(d["config"].get("prune_policy"), d.get("parity"))
PY
tail -3 $O/bench.err
