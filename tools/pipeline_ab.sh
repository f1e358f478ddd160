#!/bin/bash
# tools/pipeline_ab.sh -- on the GPU box: the bench headline at other pipeline shapes
# (calls in flight x batches per call; shipped 2 x 8), shipped library, two runs each
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
for rep in 1 2; do
  for cfg in "2 8" "3 8" "2 16" "3 4" "4 4" "2 12"; do
    set -- $cfg
    timeout 300 python bench.py --no-cpu --headline-only --min-seconds 3 --inflight $1 --steps-per-call $2 > /tmp/ab.json 2> /tmp/ab.err
    rc=$?
    if [ $rc -eq 0 ]; then
      python - "$1 in flight x $2 batches" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1])
print("%-26s %.2f M  ms/step %.3f  launch %.1f ms  MB/call %.0f" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"],
      d["roofline"].get("kernel_ms", 0) or 0, d["config"].get("device_mb_per_call", 0)))
PY
    else echo "$cfg rc=$rc"; tail -3 /tmp/ab.err; fi
  done
done
