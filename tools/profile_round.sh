#!/bin/bash
# tools/profile_round.sh rNN -- on the GPU box (through gpurun): the rocprofv3 kernel trace
# of the bench command (+ any further arguments: `tools/profile_round.sh r06_min200 --min-active 200`), the two PMC passes the HBM-traffic figure needs (FETCH_SIZE and
# WRITE_SIZE do not fit one pass), then the round's bench line.  Back in the container:
#   python tools/summarize_profile.py rNN; cp gpurun_out/bench_rNN.json profiles/rNN_bench_line.json
R=${1:-r02}
EXTRA="${@:2}"
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
mkdir -p gpurun_out
# (which library these numbers are of: tools/check_profiles.py compares it with the shipped one)
sha256sum "$ROOT"/kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip.so | cut -c1-16 > "$ROOT"/gpurun_out/prof_${R}_lib.sha256
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT"/gpurun_out/prof_${R}_trace "$ROOT"/gpurun_out/prof_${R}_fetch "$ROOT"/gpurun_out/prof_${R}_write
rocprofv3 --kernel-trace --stats -d "$ROOT"/gpurun_out/prof_${R}_trace -o runc -f csv -- python3 "$ROOT"/bench.py --no-cpu --headline-only $EXTRA > "$ROOT"/gpurun_out/prof_${R}_trace.json 2> "$ROOT"/gpurun_out/prof_${R}_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$ROOT"/gpurun_out/prof_${R}_fetch -o runc -f csv -- python3 "$ROOT"/bench.py --no-cpu --headline-only $EXTRA > /dev/null 2> "$ROOT"/gpurun_out/prof_${R}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$ROOT"/gpurun_out/prof_${R}_write -o runc -f csv -- python3 "$ROOT"/bench.py --no-cpu --headline-only $EXTRA > /dev/null 2> "$ROOT"/gpurun_out/prof_${R}_write.err
# the summary is made here too, so that the bench line below takes `roofline.traffic` from
# THIS round's counter passes (bench.py reads profiles/rNN_summary.json); the same
# summary is made again in the container from the merged gpurun_out/
cd "$ROOT"
python3 tools/summarize_profile.py $R > /dev/null
if [ -z "$EXTRA" ]; then python3 bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err; fi
ls "$ROOT"/gpurun_out/prof_${R}_trace/* | head
