#!/bin/bash
# VGPR / SGPR / LDS / scratch of every kernel of the library (code-object metadata of csrc/build/*.o; no GPU needed):
#   tools/kernel_resources.sh [name pattern]
HERE=$(cd "$(dirname "$0")" && pwd)
LLVM=/opt/rocm/lib/llvm/bin
TMP=$(mktemp -d)
for o in "$HERE"/../hair-gs_amd/csrc/build/*.o; do
  b=$(basename "$o" .o)
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin "$o" "$TMP/$b.fat" 2>/dev/null || continue
  $LLVM/clang-offload-bundler --unbundle --type=o --input="$TMP/$b.fat" --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output="$TMP/$b.co" 2>/dev/null || continue
  $LLVM/llvm-readelf --notes "$TMP/$b.co" 2>/dev/null | awk '
    /\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.sgpr_count:/ {s=$2} /\.group_segment_fixed_size:/ {l=$2}
    /\.private_segment_fixed_size:/ {p=$2} /\.vgpr_spill_count:/ {sp=$2}
    /\.wavefront_size:/ {printf "%s vgpr %s sgpr %s lds %s scratch %s spill %s\n", name, v, s, l, p, sp}'
done | c++filt | sed 's/(anonymous namespace):://' | grep -E "${1:-.}" | awk '{n=$1; $1=""; printf "%-60s %s\n", substr(n,1,60), $0}'
rm -rf "$TMP"
