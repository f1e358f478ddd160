cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/ab.log
: > $out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 >> $out
for l in libkd_hip_w48.so libkd_hip_bw.so; do for w in hlg-lattice hlg-1best hl; do echo "== $l $w" >> $out; KD_HIP_LIB=$l python bench.py --no-cpu --workload $w --steps 24 >> $out 2>&1; done; done
