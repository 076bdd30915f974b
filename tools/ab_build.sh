#!/bin/bash
# Builds a variant of libcover_hip.so for same-box A/B runs (COVER_LIB_PATH=tools/ab/libcover_hip_<name>.so):
#   tools/ab_build.sh <name> <file.hip> [extra hipcc flags ...]     e.g.  tools/ab_build.sh sched1 gemm_bf16.hip -DCOVER_PC_SCHED=1
# Only <file.hip> is recompiled with the extra flags; the other objects are those of the default build (run `make` first).
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../cover_vla_amd/csrc"
mkdir -p ../../tools/ab /tmp/ab_$name
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-variable "$@" -c $src -o /tmp/ab_$name/${src%.hip}.o
objs=""
for o in gemm_bf16 gemm_v3 gemm_fp8 attention decode_attn decode_own rowops f32ops select image prof capi; do
  if [ "$o.hip" = "$src" ]; then objs="$objs /tmp/ab_$name/$o.o"; else objs="$objs $o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../../tools/ab/libcover_hip_$name.so
ls -la ../../tools/ab/libcover_hip_$name.so
