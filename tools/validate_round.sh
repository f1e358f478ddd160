#!/bin/bash
# tools/validate_round.sh rNN -- on the GPU box (through gpurun): validation of a round's final
# build -- GPU tests, minutes of fuzz (shipped and stress libraries), the large-LM workload
# three times, then the profile round (tools/profile_round.sh rNN) and the bench at other
# call shapes.
R=${1:-r03}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/fuzz_gpu.py 200 3201 2>&1 | tail -2
KD_HIP_LIB=libkd_hip_stress.so timeout 300 python tools/fuzz_gpu.py 120 3202 2>&1 | tail -2
# the global-table tier: half of the cases heavy; then the half-size-table build (make tiny)
KD_FUZZ_HEAVY=0.5 timeout 300 python tools/fuzz_gpu.py 150 3203 2>&1 | tail -1
if [ -f kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip_tiny.so ]; then
  KD_HIP_LIB=libkd_hip_tiny.so KD_FUZZ_HEAVY=0.2 timeout 300 python tools/fuzz_gpu.py 150 3204 2>&1 | tail -1
fi
for i in 1 2 3; do
  timeout 900 python bench.py --no-cpu --headline-only --workload hlg-large --min-seconds 1.0 > gpurun_out/${R}_large_$i.json 2> /tmp/b.err
  echo "hlg-large run $i rc=$? $(python -c "import json;d=json.load(open('gpurun_out/${R}_large_$i.json'));print('%.2fM kernel %.1f ms alone %.1f frac %.3f'%(d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['launch_alone']['kernel_ms'], d['roofline']['launch_alone']['frac']))" 2>/dev/null)"
done
bash tools/profile_round.sh $R 2>&1 | tail -3
python - "$R" <<'PY'
import json, sys
d=json.load(open("gpurun_out/bench_%s.json" % sys.argv[1]))
print("value %.2fM ms/step %.3f kernel_ms %.1f alone %.1f frac %.3f single %.2f ms off %.2fM traffic %s cpu %s" % (d["value"]/1e6, d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["launch_alone"]["kernel_ms"], d["roofline"]["frac"], d["single_batch_ms"], (d["value_prune_interval_off"] or 0)/1e6, d["roofline"]["traffic"], d.get("cpu_baseline",{}).get("value")))
PY
bash tools/pmc_counters.sh $R > /dev/null 2>&1; python tools/summarize_counters.py $R 2>&1 | tail -2
QUICK=1 bash tools/write_breakdown.sh 2>&1 | grep -v rocprofv3 | tail -2
for args in "--steps-per-call 12 --steps 96" "--steps-per-call 16 --steps 128" "--steps-per-call 8 --inflight 3 --steps 96"; do
  timeout 300 python bench.py --no-cpu --headline-only $args > /tmp/b.json 2>/dev/null
  echo "$args: $(python -c "import json;d=json.load(open('/tmp/b.json'));print('%.2fM kernel %.1f frac %.3f'%(d['value']/1e6, d['roofline']['kernel_ms'], d['roofline']['frac']))" 2>/dev/null)"
done
