#!/bin/bash
# Register / spill / LDS / scratch metadata of every kernel in a HIP object (code-object notes of its gfx950 bundle).
#   tools/kernel_regs.sh vsmartmom.jl_amd/csrc/vsm_strip128.o [name filter (regex)]
obj=$1; filt=${2:-.}
tmp=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy -O binary --only-section=.hip_fatbin "$obj" $tmp/fat.bin
$B/clang-offload-bundler --type=o --input=$tmp/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/dev.co --unbundle
$B/llvm-readelf --notes $tmp/dev.co | awk '
  /\.agpr_count:/ {a=$2}
  /\.group_segment_fixed_size:/ {g=$2}
  /\.name:/ {name=$2}
  /\.private_segment_fixed_size:/ {ps=$2}
  /\.vgpr_count:/ {v=$2}
  /\.vgpr_spill_count:/ {sp=$2}
  /\.wavefront_size:/ {printf "%s vgpr %s agpr %s spill %s scratch %s lds %s\n", name, v, a, sp, ps, g; a=0}
' | c++filt | grep -E "$filt"
rm -rf $tmp
