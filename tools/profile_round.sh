#!/bin/bash
# tools/profile_round.sh rNN -- on the GPU box (through gpurun): the round's bench line,
# the rocprofv3 kernel trace of the same command, and the two PMC passes the HBM-traffic
# figure needs (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Then, back in the
# container:  python tools/summarize_profile.py rNN
R=${1:-r02}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err
cd /tmp && export TMPDIR=/tmp
rm -rf "$ROOT"/gpurun_out/prof_${R}_trace "$ROOT"/gpurun_out/prof_${R}_fetch "$ROOT"/gpurun_out/prof_${R}_write
rocprofv3 --kernel-trace --stats -d "$ROOT"/gpurun_out/prof_${R}_trace -o runc -f csv -- python3 "$ROOT"/bench.py --no-cpu --headline-only > "$ROOT"/gpurun_out/prof_${R}_trace.json 2> "$ROOT"/gpurun_out/prof_${R}_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$ROOT"/gpurun_out/prof_${R}_fetch -o runc -f csv -- python3 "$ROOT"/bench.py --no-cpu --headline-only > /dev/null 2> "$ROOT"/gpurun_out/prof_${R}_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$ROOT"/gpurun_out/prof_${R}_write -o runc -f csv -- python3 "$ROOT"/bench.py --no-cpu --headline-only > /dev/null 2> "$ROOT"/gpurun_out/prof_${R}_write.err
ls "$ROOT"/gpurun_out/prof_${R}_trace/* | head
