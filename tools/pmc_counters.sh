#!/bin/bash
# tools/pmc_counters.sh rNN -- rocprofv3 --pmc passes (counters only, one group per run)
# over tools/call_timing.py 4 (calls of 512 utterances, one in flight): instruction mix,
# VALU activity, L2 hit rate of SearchKernel<1>.  Then: python tools/summarize_counters.py rNN
R=${1:-r02}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rm -rf "$ROOT"/gpurun_out/pmc_${R}_$i
  rocprofv3 --pmc $c --kernel-trace -d "$ROOT"/gpurun_out/pmc_${R}_$i -o pmc -f csv -- python3 "$ROOT"/tools/call_timing.py 4 > "$ROOT"/gpurun_out/pmc_${R}_$i.log 2>&1
done
ls "$ROOT"/gpurun_out/pmc_${R}_*/
