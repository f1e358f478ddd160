#!/bin/bash
# tools/sweep_streams.sh  -- bench value against the number of batches in flight (same box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
for s in ${@:-6 7 6 7 5}; do echo "== streams $s" >> $out; python bench.py --no-cpu --streams $s --steps 56 --arena-tokens-per-utt 2097152 --arena-links-per-utt 4194304 >> $out 2>&1; done
