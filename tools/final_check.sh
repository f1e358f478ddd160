#!/bin/bash
# tools/final_check.sh rNN -- on the GPU box, on the FINAL build of a round: GPU tests, a fuzz pass
# of every build, the profile rounds (bench workload, its min_active = 200 configuration, the
# large-LM workload), the counter passes -- then, back in the container:
#   python tools/summarize_profile.py rNN; python tools/summarize_profile.py rNN_min200
#   python tools/summarize_profile.py rNN large; python tools/summarize_counters.py rNN
#   python tools/check_profiles.py rNN      (refuses profiles of another build than the shipped one)
R=${1:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 200 python tools/fuzz_gpu.py 60 6201 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_stress.so timeout 200 python tools/fuzz_gpu.py 60 6202 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 200 python tools/fuzz_gpu.py 60 6203 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter.so timeout 200 python tools/fuzz_gpu.py 60 6205 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_tiny.so KD_FUZZ_HEAVY=0.3 timeout 200 python tools/fuzz_gpu.py 60 6204 2>&1 | tail -1
bash tools/profile_large.sh $R 2>&1 | tail -2
bash tools/profile_round.sh ${R}_min200 --min-active 200 2>&1 | tail -2
bash tools/pmc_counters.sh $R 2>&1 | tail -1
bash tools/profile_round.sh $R 2>&1 | tail -2
tail -c 600 gpurun_out/bench_$R.json
