#!/bin/bash
# tools/final_check.sh rNN -- on the GPU box: the short form of tools/validate_round.sh for a
# late kernel change -- GPU tests, a fuzz pass of every build, then both profile rounds.
R=${1:-r05}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/fuzz_gpu.py 50 6201 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_stress.so timeout 200 python tools/fuzz_gpu.py 40 6202 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 200 python tools/fuzz_gpu.py 30 6203 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_tiny.so KD_FUZZ_HEAVY=0.3 timeout 200 python tools/fuzz_gpu.py 40 6204 2>&1 | tail -1
bash tools/profile_large.sh $R 2>&1 | tail -2
bash tools/profile_round.sh $R 2>&1 | tail -2
tail -c 600 gpurun_out/bench_$R.json
