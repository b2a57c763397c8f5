#!/bin/bash
# per-kernel register / LDS / spill counts of one built object of tvts_amd/csrc/build (the gfx950 code object's metadata)
# usage: tools/kernel_regs.sh norm.o [name filter]
set -e
O=$(dirname "$0")/../tvts_amd/csrc/build/$1
L=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat "$O"
$L/clang-offload-bundler --unbundle --input=$T/fat --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.o
$L/llvm-readelf --notes $T/dev.o | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|\.group_segment_fixed_size|vgpr_spill_count|sgpr_spill" | paste - - - - - - | sed 's/ \+/ /g' | grep -E "${2:-.}" | awk '{print}' 
rm -rf $T
