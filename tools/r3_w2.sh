#!/bin/bash
# round-3: GPU tests + bench at several call sizes + phase profile
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=${1:-w2}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_${T}_tests.txt 2>&1
tail -3 gpurun_out/r3_${T}_tests.txt
for spc in 8 24 32; do
  timeout 900 python bench.py --no-cpu --headline-only --steps-per-call $spc --steps $((spc*8)) > gpurun_out/r3_${T}_bench_$spc.json 2> gpurun_out/r3_${T}_bench_$spc.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r3_${T}_bench_$spc.json"))
    print("spc $spc value %.2fM ms/step %.3f kernel_ms %.1f alone_ms %.1f frac %.3f alone_frac %.3f single %.2f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["launch_alone"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["launch_alone"]["frac"], d["single_batch_ms"]))
except Exception as e:
    print("spc $spc failed", e)
PY
done
KD_HIP_LIB=libkd_hip_prof.so timeout 600 python tools/prof_phases.py hlg-lattice > gpurun_out/r3_${T}_prof.txt 2>&1
KD_HIP_LIB=libkd_hip_prof.so timeout 600 python tools/prof_phases.py hlg-lattice 1536 > gpurun_out/r3_${T}_prof1536.txt 2>&1
grep -v "^/opt" gpurun_out/r3_${T}_prof1536.txt | head -30
