#!/bin/bash
# tools/profile_large.sh rNN -- on the GPU box: the rocprofv3 kernel trace of the large-LM
# workload's bench command (BASELINE configs[4] on one GPU) and its bench line
R=${1:-r03}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$ROOT"/gpurun_out
# (which library these numbers are of: tools/check_profiles.py compares it with the shipped one)
sha256sum "$ROOT"/kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip.so | cut -c1-16 > "$ROOT"/gpurun_out/prof_${R}_large_lib.sha256
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT"/gpurun_out/prof_${R}_large
rocprofv3 --kernel-trace --stats -d "$ROOT"/gpurun_out/prof_${R}_large -o runc -f csv -- python3 "$ROOT"/bench.py --workload hlg-large --no-cpu --headline-only > "$ROOT"/gpurun_out/prof_${R}_large.json 2> "$ROOT"/gpurun_out/prof_${R}_large.err
# the two PMC passes of the HBM-traffic figure (FETCH_SIZE and WRITE_SIZE do not fit one pass;
# counters only, with --kernel-trace: MI355X_MICROARCH.md) -- this is the one workload whose
# graph (2.26 GB of arcs) is nine times the Infinity Cache, where "HBM bytes" is about HBM
rm -rf "$ROOT"/gpurun_out/prof_${R}_large_fetch "$ROOT"/gpurun_out/prof_${R}_large_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$ROOT"/gpurun_out/prof_${R}_large_fetch -o runc -f csv -- python3 "$ROOT"/bench.py --workload hlg-large --no-cpu --headline-only --min-seconds 3 > /dev/null 2> "$ROOT"/gpurun_out/prof_${R}_large_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$ROOT"/gpurun_out/prof_${R}_large_write -o runc -f csv -- python3 "$ROOT"/bench.py --workload hlg-large --no-cpu --headline-only --min-seconds 3 > /dev/null 2> "$ROOT"/gpurun_out/prof_${R}_large_write.err
ls "$ROOT"/gpurun_out/prof_${R}_large/
tail -1 "$ROOT"/gpurun_out/prof_${R}_large.json | cut -c1-400
