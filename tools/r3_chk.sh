#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
T=${1:-chk}
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3_${T}_tests.txt 2>&1
tail -3 gpurun_out/r3_${T}_tests.txt
timeout 900 python bench.py --no-cpu --headline-only > gpurun_out/r3_${T}_bench.json 2> gpurun_out/r3_${T}_bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/r3_${T}_bench.json"))
print("value %.2fM ms/step %.3f kernel_ms %.1f alone_ms %.1f frac %.3f alone_frac %.3f single %.2f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["launch_alone"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["launch_alone"]["frac"], d["single_batch_ms"]))
PY
