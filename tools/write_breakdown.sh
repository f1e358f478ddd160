#!/bin/bash
# tools/write_breakdown.sh [lib...] -- on the GPU box: HBM write bytes per launch (rocprofv3 --pmc
# WRITE_SIZE, counters only) of calls of 512 utterances: the bench workload at prune_interval
# 25 and off, FasterDecoder on the same graph; then the same for side builds
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
run() {  # tag lib workload interval counters
  rm -rf "$ROOT"/gpurun_out/wb_$1
  KD_HIP_LIB=$2 rocprofv3 --pmc $5 --kernel-trace -d "$ROOT"/gpurun_out/wb_$1 -o pmc -f csv -- python3 "$ROOT"/tools/write_probe.py $3 $4 4 $6 > "$ROOT"/gpurun_out/wb_$1.log 2>&1
  python3 - "$1" "$ROOT"/gpurun_out/wb_$1 <<'PY'
import csv, glob, sys, collections
tag, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "SearchKernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(tag, {k: "%.3f G (n=%d)" % (sum(v) / len(v) / 1e6 * (1.024 if k.endswith("_SIZE") else 1e-3), len(v)) for k, v in acc.items()})
PY
  grep "per utt" "$ROOT"/gpurun_out/wb_$1.log
}
for lib in "" "$@"; do
  l=libkd_hip${lib:+_$lib}.so
  run lat25_$lib $l hlg-lattice 25 WRITE_SIZE
  if [ -z "$QUICK" ]; then
    run latoff_$lib $l hlg-lattice 1000 WRITE_SIZE
    run best_$lib $l hlg-1best 0 WRITE_SIZE
    for b in 8 10 12 13 14; do run lat25_b${b}_$lib $l hlg-lattice 25 WRITE_SIZE $b; done
  fi
done
rocprofv3 --list-avail 2>/dev/null | grep -i -o "TCC_[A-Z0-9_]*\(WR\|ATOMIC\|WRITEBACK\)[A-Za-z0-9_\[\]]*" | sort -u | head -60 > "$ROOT"/gpurun_out/wb_counters.txt
