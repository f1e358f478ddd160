#!/bin/bash
# round 6, eighth GPU session: split tests with `stay`; knob builds on the register allocation of this round
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step8
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -4
timeout 200 python tools/fuzz_gpu.py 60 8201 2>&1 | tail -1
KD_HIP_LIB=libkd_hip_jitter.so timeout 200 python tools/fuzz_gpu.py 60 8202 2>&1 | tail -1
bench() { # name env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --headline-only --min-seconds 3 > $O/$n.json 2> $O/$n.err
  rc=$?
  if [ $rc -eq 0 ]; then python - $n $O/$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s %.2f M  ms/step %.3f  launch %.1f ms  alone %.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("kernel_ms", 0) or 0, d["roofline"]["launch_alone"]["kernel_ms"]), flush=True)
PY
  else echo "$n rc=$rc"; tail -5 $O/$n.err; fi
}
for rep in 1 2; do
bench shipped
bench b1024 KD_HIP_LIB=libkd_hip_b1024.so
bench g2 KD_HIP_LIB=libkd_hip_g2.so
bench sd8 KD_HIP_LIB=libkd_hip_sd8.so
bench ilp8 KD_HIP_LIB=libkd_hip_ilp8.so
done
lbench() { n=$1; shift; env "$@" timeout 400 python bench.py --no-cpu --headline-only --workload hlg-large --min-seconds 3 > $O/L$n.json 2> $O/L$n.err && python - L$n $O/L$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s %.2f M  ms/step %.3f  launch %.1f ms  alone %.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("kernel_ms", 0) or 0, d["roofline"]["launch_alone"]["kernel_ms"]), flush=True)
PY
}
lbench shipped
lbench b1024 KD_HIP_LIB=libkd_hip_b1024.so
lbench ilp8 KD_HIP_LIB=libkd_hip_ilp8.so
