#!/bin/bash
# tools/ab_step.sh libA.so libB.so ...  same-box A/B of whole steps: per-piece host timing (tools/step_timing.py),
# the bench with six batches in flight, then the GPU test suite with the default library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
for l in "$@"; do echo "== $l" >> $out; KD_HIP_LIB=$l python tools/step_timing.py >> $out 2>&1; KD_HIP_LIB=$l python bench.py --no-cpu --streams 6 --steps 48 >> $out 2>&1; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $out
