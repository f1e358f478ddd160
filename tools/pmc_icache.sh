#!/bin/bash
# tools/pmc_icache.sh -- on the GPU box: instruction-cache counters of SearchKernel<1> over
# tools/call_timing.py 4 (counters only, one group per run)
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_DCACHE_MISSES"; do
  i=$((i+1))
  rm -rf "$ROOT"/gpurun_out/pmc_ic_$i
  rocprofv3 --pmc $c --kernel-trace -d "$ROOT"/gpurun_out/pmc_ic_$i -o pmc -f csv -- python3 "$ROOT"/tools/call_timing.py 4 > "$ROOT"/gpurun_out/pmc_ic_$i.log 2>&1 || tail -3 "$ROOT"/gpurun_out/pmc_ic_$i.log
done
python3 - "$ROOT" <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for d in sorted(glob.glob(root + "/gpurun_out/pmc_ic_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if "SearchKernel" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for k in acc:
            print("%-34s %16.0f per launch (%d launches)" % (k, acc[k] / max(n[k], 1), n[k]))
PY
