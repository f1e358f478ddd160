#!/bin/bash
# tools/ab_workloads.sh libA.so libB.so ...  same-box A/B over the three bench workloads
# (hlg-lattice, hlg-1best, hl), after the GPU test suite with the default library.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $out
for l in "${@:-libkd_hip.so}"; do for w in hlg-lattice hlg-1best hl; do echo "== $l $w" >> $out; KD_HIP_LIB=$l python bench.py --no-cpu --workload $w --steps 28 >> $out 2>&1; done; done
