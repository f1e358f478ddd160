#!/bin/bash
# Registers, spills, scratch and LDS of every kernel in a built library, read from the
# code object embedded in it (its .hip_fatbin section): tools/kernel_meta.sh [lib.so]
LIB=$(readlink -f "${1:-$(dirname "$0")/../kaldi-decoder_amd/python/kaldi_decoder/lib/libkd_hip.so}")
LLVM=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$T/fatbin" || exit 1
$LLVM/clang-offload-bundler --unbundle --type=o --input="$T/fatbin" \
  --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$T/kd.co" 2>/dev/null || \
$LLVM/clang-offload-bundler --unbundle --type=o --input="$T/fatbin" \
  --targets=hip-amdgcn-amd-amdhsa--gfx950 --output="$T/kd.co" || exit 1
$LLVM/llvm-readelf --notes "$T/kd.co" | awk '
  /\.name:/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.vgpr_spill_count:/ {vs=$2} /\.sgpr_spill_count:/ {ss=$2}
  /\.private_segment_fixed_size:/ {pr=$2} /\.group_segment_fixed_size:/ {l=$2}
  /\.agpr_count:/ {a=$2}
  /\.wavefront_size:/ {printf "%-58s vgpr %3d agpr %3d sgpr %3d  spill v %3d s %3d  scratch %4d B  lds %6d B\n", name, v, a, s, vs, ss, pr, l}'
rm -rf "$T"
