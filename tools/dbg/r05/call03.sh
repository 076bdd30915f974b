# round 5, GPU call 3: in-kernel probe of gemm_v3 (prologue / loop / epilogue of workgroup 0, cycles per k-tile, loop clock)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for r in 0 23; do echo "== COVER_V3_RING=$r M=448"; COVER_V3_RING=$r timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call03_m448.txt
echo "== COVER_TILE_PICK=o M=448"; COVER_TILE_PICK=o timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/call03_m448.txt
echo "== pi0 M=2232 pick n"; SHAPES=pi0 COVER_TILE_PICK=n timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call03_pi0.txt
