#!/bin/bash
# tools/pmc_pass.sh "<counters>" ...   one rocprofv3 --pmc pass per argument over tools/step_timing.py
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rm -rf "$ROOT"/gpurun_out/pmc_$i
  rocprofv3 --pmc $c --kernel-trace -d "$ROOT"/gpurun_out/pmc_$i -o pmc -f csv -- python3 "$ROOT"/tools/step_timing.py > "$ROOT"/gpurun_out/pmc_$i.log 2>&1
done
