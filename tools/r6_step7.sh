#!/bin/bash
# round 6, seventh GPU session: the whole GPU suite on the build whose calls are split only when asked,
# fuzz passes of every build (two cases in three split, random thresholds), the large-LM bench, the bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step7
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 300 python tools/fuzz_gpu.py 90 7201 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_stress.so timeout 300 python tools/fuzz_gpu.py 90 7202 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 300 python tools/fuzz_gpu.py 90 7203 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_jitter.so timeout 300 python tools/fuzz_gpu.py 90 7205 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_tiny.so KD_FUZZ_HEAVY=0.3 timeout 300 python tools/fuzz_gpu.py 90 7204 2>&1 | tail -2
timeout 600 python bench.py --no-cpu --headline-only --workload hlg-large --min-seconds 4 > $O/large.json 2> $O/large.err; tail -c 400 $O/large.json; echo
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json; echo
