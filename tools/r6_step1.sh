#!/bin/bash
# round 6, first GPU session: the two-kernel launch -- unsplit path, forced hand-overs, bench A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step1
mkdir -p $O
T="tests/test_gpu_lattice.py tests/test_gpu_faster.py tests/test_gpu_golden.py tests/test_kat.py tests/test_kat_deep.py"
echo "== unsplit (small batches)"; timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -5
echo "== every call split, hand-overs at 3 / 6 tokens"
KD_SPLIT_MIN_BATCH=1 KD_MIG_HI=6 KD_MIG_LO=3 timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -15
echo "== every call split, default thresholds, pruning on the small kind"
KD_SPLIT_MIN_BATCH=1 KD_PRUNE_SMALL=1 timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -15
bench() { # name env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --headline-only --min-seconds 3 > $O/$n.json 2> $O/$n.err
  rc=$?
  if [ $rc -eq 0 ]; then python - $n $O/$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s %.2f M  ms/step %.3f  launch %.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("kernel_ms", 0) or 0))
PY
  else echo "$n rc=$rc"; tail -5 $O/$n.err; fi
}
for rep in 1 2; do
  bench r5 KD_HIP_LIB=libkd_hip_r5.so
  bench new_unsplit KD_SPLIT=0
  bench new_split
  bench new_split_lo128 KD_MIG_LO=128
  bench new_split_prune KD_PRUNE_SMALL=1
done
