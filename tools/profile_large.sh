#!/bin/bash
# tools/profile_large.sh rNN -- on the GPU box: the rocprofv3 kernel trace of the large-LM
# workload's bench command (BASELINE configs[4] on one GPU) and its bench line
R=${1:-r03}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT"/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT"/gpurun_out/prof_${R}_large
rocprofv3 --kernel-trace --stats -d "$ROOT"/gpurun_out/prof_${R}_large -o runc -f csv -- python3 "$ROOT"/bench.py --workload hlg-large --no-cpu --headline-only > "$ROOT"/gpurun_out/prof_${R}_large.json 2> "$ROOT"/gpurun_out/prof_${R}_large.err
ls "$ROOT"/gpurun_out/prof_${R}_large/
tail -1 "$ROOT"/gpurun_out/prof_${R}_large.json | cut -c1-400
