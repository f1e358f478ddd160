#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
O=gpurun_out/r6_step6
mkdir -p $O
echo "== split test, traceback"; timeout 600 python -m pytest tests/test_gpu_split.py -x -q -m gpu 2>&1 | tail -40
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
bench x224_32_64 KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=32 KD_MIG_HI=64
bench x224_32_64_share1 KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=32 KD_MIG_HI=64 KD_SPLIT_SHARERS=1
bench x208_32_128 KD_SPLIT_LARGE_CUS=208 KD_MIG_LO=32 KD_MIG_HI=128
bench x192_32_256 KD_MIG_LO=32
bench x224_8_32 KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=8 KD_MIG_HI=32
bench s128_x224_8_32 KD_HIP_LIB=libkd_hip_s128.so KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=8 KD_MIG_HI=32
bench s128_x224_32_128 KD_HIP_LIB=libkd_hip_s128.so KD_SPLIT_LARGE_CUS=224 KD_MIG_LO=32 KD_MIG_HI=128
echo "== probe s128"; KD_HIP_LIB=libkd_hip_s128.so KD_SPLIT_LARGE_CUS=224 timeout 300 python tools/split_probe.py 1024 off:off a:8:32:0 b:32:128:0 2>&1 | grep -v amdgpu.ids | tail -12
