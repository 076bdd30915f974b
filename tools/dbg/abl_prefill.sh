#!/bin/bash
# Ablation of the loader-wave tiled GEMM (gemm_tiled_pc) at the prefill shapes: builds copies of the library with -DCOVER_PC_ABL=<bits>
# (HERE, hipcc cross-compiles) as tools/ab/libcover_hip_abl<bits>.so; on the GPU box:
#   for a in 1 2 4 8 12 3; do COVER_LIB_PATH=tools/ab/libcover_hip_abl$a.so python tools/dbg/bench_prefill.py 448 3; done
# bits: 1 no MFMAs, 2 no LDS fragment reads, 4 no weight DMA, 8 no activation DMA (results are garbage by design).
cd "$(dirname "$0")/../../cover_vla_amd/csrc"
mkdir -p ../../tools/ab
for a in "$@"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable -DCOVER_PC_ABL=$a -c gemm_bf16.hip -o /tmp/gemm_bf16_abl$a.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC /tmp/gemm_bf16_abl$a.o gemm_fp8.o attention.o decode_attn.o decode_own.o rowops.o f32ops.o select.o image.o prof.o capi.o -o ../../tools/ab/libcover_hip_abl$a.so ) 2>&1 | grep -E "error|Illegal" &
done
wait
ls -la ../../tools/ab/
