#!/bin/bash
# tools/ab_libs.sh name... -- on the GPU box: the bench headline for the shipped library and
# for side builds libkd_hip_<name>.so (make knob NAME=<name> KNOBS=...), two runs each,
# interleaved so that box drift shows
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
for rep in 1 2; do
  for n in "" "$@"; do
    lib=libkd_hip${n:+_$n}.so
    KD_HIP_LIB=$lib timeout 300 python bench.py --no-cpu --headline-only --min-seconds 3 > /tmp/ab.json 2> /tmp/ab.err
    rc=$?
    if [ $rc -eq 0 ]; then
      python - "$lib" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("%-22s %.2f M  ms/step %.3f  launch %.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"],
      d["roofline"].get("kernel_ms", 0) or 0))
PY
    else echo "$lib rc=$rc"; tail -3 /tmp/ab.err; fi
  done
done
