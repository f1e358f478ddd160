#!/bin/bash
# tools/pmc_counters.sh rNN -- rocprofv3 --pmc passes (counters only, one group per run)
# over tools/call_timing.py 4 (calls of 512 utterances, one in flight): instruction mix, where
# the wave-cycles go (issuing / waiting on a counter / neither), lane utilisation, LDS
# conflicts, average vector-memory and LDS latency (INST_LEVEL / INSTS), L2 hit rate of
# SearchKernel<1>.  Then: python tools/summarize_counters.py rNN
R=${1:-r02}
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
# (which library these numbers are of: tools/check_profiles.py compares it with the shipped one)
sha256sum "$ROOT"/kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip.so | cut -c1-16 > "$ROOT"/gpurun_out/pmc_${R}_lib.sha256
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" \
         "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES" \
         "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
         "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU" \
         "SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
         "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INSTS_LDS_ATOMIC" \
         "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
         "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_CYCLES"; do
  i=$((i+1))
  rm -rf "$ROOT"/gpurun_out/pmc_${R}_$i
  rocprofv3 --pmc $c --kernel-trace -d "$ROOT"/gpurun_out/pmc_${R}_$i -o pmc -f csv -- python3 "$ROOT"/tools/call_timing.py 4 > "$ROOT"/gpurun_out/pmc_${R}_$i.log 2>&1
done
ls "$ROOT"/gpurun_out/pmc_${R}_*/
