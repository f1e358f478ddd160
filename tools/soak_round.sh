#!/bin/bash
# tools/soak_round.sh -- on the GPU box, on a round's final build: what the driver runs at round end
# (smoke, the default bench line, timed), the bench over and over, the bench-scale parity tests with
# every call dealt out over two kinds of workgroups and with the timing-perturbation library in place
# of the shipped one, minutes of fuzz on the jitter builds with a share of heavy cases.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ulimit -c 0
mkdir -p gpurun_out
t0=$(date +%s); timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1; echo "smoke $(( $(date +%s) - t0 )) s"
t0=$(date +%s); timeout 900 python bench.py > gpurun_out/soak_bench.json 2> gpurun_out/soak_bench.err; echo "default bench rc=$? $(( $(date +%s) - t0 )) s: $(python -c "import json;d=json.loads(open('gpurun_out/soak_bench.json').read().strip().splitlines()[-1]);print('%.2f M, eager %.2f, large %.2f' % (d['value']/1e6, d['value_prune_eager']/1e6, d['large_lm']['value']/1e6))")"
bash tools/soak.sh 10 hlg-lattice 2>&1 | tail -3
bash tools/soak.sh 3 hlg-large 2>&1 | tail -3
KD_SPLIT=1 timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "bench_scale" 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_lazy_prune.py tests/test_gpu_split.py -q -m gpu -k "bench_scale or cfg5 or cfg2 or lazy or split or prune or hands" 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_jitter_full.so timeout 200 python tools/fuzz_gpu.py 120 4101 2>&1 | tail -1
KD_FUZZ_HEAVY=0.4 KD_HIP_LIB=libkd_hip_jitter.so timeout 200 python tools/fuzz_gpu.py 120 4102 2>&1 | tail -1
KD_FUZZ_HEAVY=0.4 timeout 200 python tools/fuzz_gpu.py 120 4103 2>&1 | tail -1
for w in hlg-lattice hlg-large; do
  KD_HIP_LIB=libkd_hip_jitter_full.so timeout 900 python bench.py --no-cpu --headline-only --workload $w --min-seconds 0.6 > /tmp/j.json 2> /tmp/j.err
  echo "jitter bench $w rc=$?"
done
