#!/bin/bash
# Device ISA of one object file built by hipcc: tools/disasm.sh <file.o> <out.s>   (then grep the mangled kernel name)
set -e
t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-objdump -d $t/dev.co > "$2"
rm -rf $t
