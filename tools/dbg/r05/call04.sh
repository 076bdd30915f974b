# round 5, GPU call 4: gemm_v3 with one wave per SIMD (4 waves: 224x96 of 112x48, 224x192 of 112x96, 224x128 of 112x64): parity of forced picks, then A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for p in r s t n o; do echo "== fuzz COVER_TILE_PICK=$p"; COVER_TILE_PICK=$p timeout 600 python tools/dbg/fuzz_gemm.py 150 7 2>&1 | tail -3; done
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2
for p in r s t; do echo "== COVER_TILE_PICK=$p M=448"; COVER_TILE_PICK=$p timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call04_m448.txt
for p in r t; do echo "== COVER_TILE_PICK=$p COVER_V3_RING=34 M=448"; COVER_V3_RING=34 COVER_TILE_PICK=$p timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/call04_m448.txt
for p in s t; do echo "== pi0 M=2232 pick $p"; SHAPES=pi0 COVER_TILE_PICK=$p timeout 300 python tools/dbg/bench_prefill.py 2232 3; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call04_pi0.txt
