#!/bin/bash
# Register / scratch / LDS use of every kernel in an object file built by hipcc: tools/kres.sh <file.o> [name-substring]
set -e
t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin "$1"
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/dev.co --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for blk in txt.split('- .agpr_count:')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if flt in name:
        import subprocess
        dn=subprocess.run(['c++filt',name],capture_output=True,text=True).stdout.strip()[:90]
        print(f\"vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>4} sgpr {g('sgpr_count'):>4} scratch {g('private_segment_fixed_size'):>5} spill_v {g('vgpr_spill_count'):>4}  {dn}\")
" "$2"
rm -rf $t
