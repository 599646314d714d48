#!/bin/bash
# usage: tools/kernel_regs.sh psvo_amd/csrc/bsim_bwd_dx2.o   -> VGPR / AGPR / spill / LDS figures of every kernel in the object
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d /tmp/kregs.XXXXXX)
$B/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fatbin
tgt=$($B/clang-offload-bundler --type=o --input=$T/fatbin --list | grep gfx950)
$B/clang-offload-bundler --type=o --targets=$tgt --input=$T/fatbin --output=$T/co --unbundle
$B/llvm-readelf --notes $T/co | grep -E "^\s+\.name:|\.vgpr_count|\.vgpr_spill_count|\.agpr_count|group_segment_fixed" | paste - - - - - | sed 's/ \+/ /g'
