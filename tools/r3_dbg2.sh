#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
for i in 1 2; do
timeout 600 python bench.py --no-cpu --headline-only > gpurun_out/r3_full_bench.json 2> gpurun_out/r3_full_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3_full_bench.json"))
print("value %.2fM ms/step %.3f kernel_ms %.1f alone_ms %.1f frac %.3f single %.2f ms" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["launch_alone"]["kernel_ms"], d["roofline"]["frac"], d["single_batch_ms"]))
PY
done
tail -3 gpurun_out/r3_full_bench.err
