#!/bin/bash
# tools/sweep_queues.sh -- bench value against (batches in flight, GPU_MAX_HW_QUEUES) pairs, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
for sq in "7 8" "8 10" "9 12" "10 12" "7 8"; do set -- $sq; echo "== streams $1 queues $2" >> $out; GPU_MAX_HW_QUEUES=$2 python bench.py --no-cpu --streams $1 --steps $((8*$1)) >> $out 2>&1; done
