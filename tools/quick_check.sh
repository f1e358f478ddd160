#!/bin/bash
# tools/quick_check.sh [side build names] -- on the GPU box: quick fuzz of every build, the bench-scale parity tests,
# then an A/B of the side builds named on the command line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
FZ=${FUZZ_S:-45}
timeout 300 python tools/fuzz_gpu.py $FZ 5101 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_stress.so timeout 300 python tools/fuzz_gpu.py $FZ 5102 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_jitter.so timeout 300 python tools/fuzz_gpu.py $FZ 5103 2>&1 | tail -2
if [ -f kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip_tiny.so ]; then
  KD_HIP_LIB=libkd_hip_tiny.so KD_FUZZ_HEAVY=0.3 timeout 300 python tools/fuzz_gpu.py $FZ 5104 2>&1 | tail -2
fi
KD_FUZZ_HEAVY=0.5 timeout 300 python tools/fuzz_gpu.py $FZ 5105 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "${TESTS_K:-bench_scale_hlg or cfg3 or cfg2_full}" 2>&1 | tail -5
bash tools/ab_libs.sh "$@" 2>&1 | tail -20
