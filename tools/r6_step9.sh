#!/bin/bash
# round 6, ninth GPU session: PruneActiveTokens owed or run -- tests, fuzz, the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step9
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lazy_prune.py tests/test_gpu_lattice.py tests/test_kat_deep.py tests/test_kat_unpinned.py -x -q -m gpu 2>&1 | tail -15
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 200 python tools/fuzz_gpu.py 60 9201 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_stress.so timeout 200 python tools/fuzz_gpu.py 60 9202 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter.so timeout 200 python tools/fuzz_gpu.py 60 9203 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_tiny.so KD_FUZZ_HEAVY=0.3 timeout 200 python tools/fuzz_gpu.py 60 9204 2>&1 | tail -1
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
