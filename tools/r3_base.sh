#!/bin/bash
# round-3 baseline: GPU tests, phase profiles (default / tier-1 detail / tail detail), frame histogram
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-base}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_${T}_tests.txt 2>&1
KD_HIP_LIB=libkd_hip_prof_t1.so timeout 600 python tools/prof_phases.py hlg-lattice > gpurun_out/r3_${T}_prof_t1.txt 2>&1
KD_HIP_LIB=libkd_hip_prof.so timeout 600 python tools/prof_phases.py hlg-lattice > gpurun_out/r3_${T}_prof.txt 2>&1
timeout 600 python tools/frame_hist.py > gpurun_out/r3_${T}_hist.txt 2>&1
timeout 900 python bench.py --no-cpu --headline-only > gpurun_out/r3_${T}_bench.json 2> gpurun_out/r3_${T}_bench.err
tail -3 gpurun_out/r3_${T}_tests.txt; cat gpurun_out/r3_${T}_bench.json | cut -c1-600
