cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
for l in libkd_hip_w2.so libkd_hip_w2b.so; do for s in 1 4 6 8; do echo "== $l streams $s" >> $out; KD_HIP_LIB=$l python bench.py --no-cpu --streams $s --steps 24 --arena-tokens-per-utt 1048576 --arena-links-per-utt 2097152 >> $out 2>&1; done; done
