#!/bin/bash
# tools/jitter_soak.sh -- on the GPU box (through gpurun): the timing-perturbation builds
# (make jitter) against what the shipped build is validated with -- minutes of fuzz on both
# jitter libraries (shipped and shrunk tail capacities, a share of the cases in the
# global-table tier), the bench-scale parity tests, and a few bench runs of both workloads.
# A wave may lag any other by thousands of cycles behind every barrier there: results must
# not change, nothing may hang.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 400 python tools/fuzz_gpu.py 300 4101 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter.so timeout 400 python tools/fuzz_gpu.py 300 4102 2>&1 | tail -1
KD_FUZZ_HEAVY=0.5 KD_HIP_LIB=libkd_hip_jitter_full.so timeout 300 python tools/fuzz_gpu.py 200 4103 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 1500 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "bench_scale or cfg5 or cfg2" 2>&1 | tail -2
for w in hlg-lattice hlg-large; do
  for i in 1 2 3; do
    KD_HIP_LIB=libkd_hip_jitter_full.so timeout 900 python bench.py --no-cpu --headline-only --workload $w --min-seconds 0.6 > /tmp/j.json 2> /tmp/j.err
    echo "jitter bench $w run $i rc=$? $(python -c "import json;d=json.load(open('/tmp/j.json'));print('%.2fM (not a timing: the build sleeps)'%(d['value']/1e6))" 2>/dev/null)"
  done
done
