#!/bin/bash
# tools/fuzz_round.sh seconds seed -- on the GPU box: fuzz over the shipped, stress, tiny and both
# jitter builds, `seconds` each (pools that are not to grow among the cases: tools/fuzz_gpu.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
S=${1:-120}; SEED=${2:-6000}
i=0
for lib in libkd_hip.so libkd_hip_stress.so libkd_hip_tiny.so libkd_hip_jitter.so libkd_hip_jitter_full.so; do
  i=$((i+1))
  echo "== $lib"
  KD_HIP_LIB=$lib timeout $((S + 120)) python tools/fuzz_gpu.py $S $((SEED + i)) 2>&1 | grep -v amdgpu.ids | tail -2
done
