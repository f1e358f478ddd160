#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
mkdir -p gpurun_out/q
timeout 900 python -m pytest tests/test_gpu_lazy_prune.py tests/test_gpu_lattice.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/min200_probe.py 128 2>&1 | grep -v amdgpu.ids | tail -3
timeout 900 python bench.py > gpurun_out/q/bench.json 2> gpurun_out/q/bench.err; python - gpurun_out/q/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.2f M  eager %.2f  off %.2f  min200 %.2f  large %s  single %.2f ms  launch %.1f alone %.1f  MB/call %.0f" % (
    d["value"] / 1e6, (d.get("value_prune_eager") or 0) / 1e6, (d.get("value_prune_interval_off") or 0) / 1e6,
    (d.get("value_min_active_200") or 0) / 1e6, (d.get("large_lm") or {}).get("value"), d["single_batch_ms"],
    d["roofline"]["kernel_ms"], d["roofline"]["launch_alone"]["kernel_ms"], d["config"]["device_mb_per_call"]))
print(d["config"].get("prune_policy"), d.get("parity"), d["config"]["pool_grow_events"])
PY
tail -2 gpurun_out/q/bench.err
timeout 600 python bench.py --no-cpu --headline-only --workload hlg-large --min-seconds 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('large %.2f M' % (d['value']/1e6), d['config'].get('prune_policy'), d['config'].get('pool_grow_events'), 'MB/call %.0f' % d['config']['device_mb_per_call'])"
