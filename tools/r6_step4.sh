#!/bin/bash
# round 6, fourth GPU session: where the split launch's time goes; the X = 208 hang; correctness; A/B of the new flags
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step4
mkdir -p $O
echo "== probe"; timeout 600 python tools/split_probe.py 1024 2>&1 | grep -v amdgpu.ids | tail -30
echo "== probe X=208"; KD_SPLIT_LARGE_CUS=208 timeout 120 python tools/split_probe.py 1024 d256:256:256:0 2>&1 | grep -v amdgpu.ids | tail -8; echo "rc=$?"
echo "== probe X=224"; KD_SPLIT_LARGE_CUS=224 timeout 120 python tools/split_probe.py 1024 d256:256:256:0 2>&1 | grep -v amdgpu.ids | tail -8; echo "rc=$?"
echo "== probe X=160"; KD_SPLIT_LARGE_CUS=160 timeout 120 python tools/split_probe.py 1024 d256:256:256:0 2>&1 | grep -v amdgpu.ids | tail -8; echo "rc=$?"
T="tests/test_gpu_lattice.py tests/test_gpu_faster.py tests/test_gpu_golden.py tests/test_kat.py tests/test_kat_deep.py tests/test_gpu_split.py"
for lib in libkd_hip.so; do
echo "== $lib: unsplit"; KD_HIP_LIB=$lib timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -4
echo "== $lib: every call split, hand-overs at 3 / 6 tokens"
KD_HIP_LIB=$lib KD_SPLIT_MIN_BATCH=1 KD_MIG_HI=6 KD_MIG_LO=3 timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -8
echo "== $lib: every call split, default thresholds, pruning on the small kind"
KD_HIP_LIB=$lib KD_SPLIT_MIN_BATCH=1 KD_PRUNE_SMALL=1 timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -8
done
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
  bench unsplit KD_SPLIT=0
  bench np_unsplit KD_SPLIT=0 KD_HIP_LIB=libkd_hip_np.so
  bench split
done
