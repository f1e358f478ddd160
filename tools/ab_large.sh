#!/bin/bash
# tools/ab_large.sh name... -- on the GPU box: the large-LM workload's headline for the shipped
# library and for side builds libkd_hip_<name>.so, two runs each, interleaved (as tools/ab_libs.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
for rep in 1 2; do
  for n in "" "$@"; do
    lib=libkd_hip${n:+_$n}.so
    KD_HIP_LIB=$lib timeout 900 python bench.py --no-cpu --headline-only --workload hlg-large --min-seconds 3 > /tmp/abl.json 2> /tmp/abl.err
    rc=$?
    if [ $rc -eq 0 ]; then
      python - "$lib" <<'PY'
import json, sys
d = json.loads(open("/tmp/abl.json").read().strip().splitlines()[-1])
print("%-22s %.2f M  launch %.1f ms  alone %.1f ms" % (sys.argv[1], d["value"] / 1e6, d["roofline"]["kernel_ms"],
      d["roofline"]["launch_alone"]["kernel_ms"]))
PY
    else echo "$lib rc=$rc"; tail -3 /tmp/abl.err; fi
  done
done
