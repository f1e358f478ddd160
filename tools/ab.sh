#!/bin/bash
# same-box A/B: tools/ab.sh libA.so libB.so ...   (each: kernel alone, six batches in flight, tier-1 phase profile if built)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
run() { echo "== $*" >> $out; env "$@" >> $out 2>&1; }
for l in "$@"; do
  run KD_HIP_LIB=$l python bench.py --no-cpu --streams 1 --steps 6
  run KD_HIP_LIB=$l python bench.py --no-cpu --streams 6 --steps 24
  p=${l%.so}_prof_t1.so
  [ -f kaldi-decoder_amd/python/kaldi_decoder/lib/$p ] && run KD_HIP_LIB=$p python tools/prof_phases.py
done
