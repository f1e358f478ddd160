#!/bin/bash
# round 6, fifth GPU session: surplus workgroups leave a call early -- the bench (two calls in flight) over the split's settings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step5
mkdir -p $O
T="tests/test_gpu_split.py tests/test_gpu_lattice.py"
echo "== split tests"; KD_SPLIT_MIN_BATCH=1 KD_MIG_HI=6 KD_MIG_LO=3 timeout 900 python -m pytest $T -x -q -m gpu 2>&1 | tail -4
bench() { # name env...
  n=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu --headline-only --min-seconds 3 > $O/$n.json 2> $O/$n.err
  rc=$?
  if [ $rc -eq 0 ]; then python - $n $O/$n.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print("%-28s %.2f M  ms/step %.3f  launch %.1f ms" % (sys.argv[1], d["value"] / 1e6, d["ms_per_step"], d["roofline"].get("kernel_ms", 0) or 0), flush=True)
PY
  else echo "$n rc=$rc"; tail -5 $O/$n.err; fi
}
bench unsplit KD_SPLIT=0
bench x192_256_256
bench x224_256_256 KD_SPLIT_LARGE_CUS=224
bench x224_32_256 KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=32
bench x224_32_64 KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=32 KD_MIG_HI=64
bench x224_8_64 KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=8 KD_MIG_HI=64
bench x208_32_256 KD_SPLIT_LARGE_CUS=208 KD_MIG_LO=32
bench x208_32_128 KD_SPLIT_LARGE_CUS=208 KD_MIG_LO=32 KD_MIG_HI=128
bench x192_32_256 KD_MIG_LO=32
bench x232_32_256 KD_SPLIT_LARGE_CUS=232 KD_MIG_LO=32
bench x240_32_128 KD_SPLIT_LARGE_CUS=240 KD_MIG_LO=32 KD_MIG_HI=128
bench x224_64_256_prune KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=64 KD_PRUNE_SMALL=1
bench unsplit2 KD_SPLIT=0
echo "== probe"; KD_SPLIT_LARGE_CUS=224 timeout 300 python tools/split_probe.py 1024 off:off a:32:256:0 b:32:64:0 2>&1 | grep -v amdgpu.ids | tail -12
