#!/bin/bash
# Instruction counts per function of a built library's gfx950 code object: ds_bpermute (wave
# shuffles through the LDS crossbar), DPP moves, s_barrier, scratch loads / stores, v_readlane:
#   tools/isa_counts.sh [lib.so]
LIB=$(readlink -f "${1:-$(dirname "$0")/../kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip.so}")
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$T/fatbin" || exit 1
$LLVM/clang-offload-bundler --unbundle --type=o --input="$T/fatbin" \
  --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/kd.co" 2>/dev/null || \
$LLVM/clang-offload-bundler --unbundle --type=o --input="$T/fatbin" \
  --targets=hip-amdgcn-amd-amdhsa--gfx950 --output="$T/kd.co" || exit 1
$LLVM/llvm-objdump -d "$T/kd.co" | awk '
  /^[0-9a-f]+ <.*>:$/ { name=$2; gsub(/[<>:]/, "", name); order[++n]=name }
  name != "" {
    tot[name]++
    if ($0 ~ /ds_bpermute|ds_permute/) bp[name]++
    if ($0 ~ /_dpp/) dpp[name]++
    if ($0 ~ /s_barrier/) bar[name]++
    if ($0 ~ /scratch_load/) sl[name]++
    if ($0 ~ /scratch_store/) ss[name]++
    if ($0 ~ /v_readlane/) rl[name]++
    if ($0 ~ /s_waitcnt/) wc[name]++
  }
  END { for (i = 1; i <= n; i++) { f = order[i]; if (tot[f] > 200)
    printf "%-70s instr %6d  bpermute %4d  dpp %4d  barrier %4d  waitcnt %5d  scratch ld %4d st %4d  readlane %5d\n",
           substr(f, 1, 70), tot[f], bp[f], dpp[f], bar[f], wc[f], sl[f], ss[f], rl[f] } }'
rm -rf "$T"
