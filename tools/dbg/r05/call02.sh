# round 5, GPU call 2: gemm_v3 with one DMA role per wave (waves 0-3 activations, 4-7 weights) and separate ring depths
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for r in 0 24 34 26; do COVER_V3_RING=$r timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2; done
for r in 0 24 23 34 26 25; do echo "== COVER_V3_RING=$r M=448"; COVER_V3_RING=$r timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call02_m448.txt
for r in 0 34 26 25; do echo "== COVER_TILE_PICK=o COVER_V3_RING=$r M=448"; COVER_TILE_PICK=o COVER_V3_RING=$r timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/call02_m448.txt
for r in 0 24 23; do echo "== COVER_TILE_PICK=n COVER_V3_RING=$r pi0 M=2232"; SHAPES=pi0 COVER_TILE_PICK=n COVER_V3_RING=$r timeout 300 python tools/dbg/bench_prefill.py 2232 3; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call02_pi0.txt
for r in 24 43; do echo "== COVER_TILE_PICK=q COVER_V3_RING=$r pi0 M=2232"; SHAPES=pi0 COVER_TILE_PICK=q COVER_V3_RING=$r timeout 300 python tools/dbg/bench_prefill.py 2232 3; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/call02_pi0.txt
