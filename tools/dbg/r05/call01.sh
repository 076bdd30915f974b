# round 5, GPU call 1: self-loading 8-wave tiled GEMM (gemm_v3.hip) -- parity of the GEMM tests, then same-box A/B against the loader-wave kernels
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -5
for v in 0 1; do echo "== COVER_V3=$v M=448"; COVER_V3=$v timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | tee gpurun_out/r05/call01_m448.txt
for p in n o; do echo "== COVER_TILE_PICK=$p M=448"; COVER_TILE_PICK=$p timeout 300 python tools/dbg/bench_prefill.py 448 4; done 2>&1 | tee -a gpurun_out/r05/call01_m448.txt
for v in 0 1; do echo "== COVER_V3=$v pi0 M=2232"; SHAPES=pi0 COVER_V3=$v timeout 300 python tools/dbg/bench_prefill.py 2232 3; done 2>&1 | tee gpurun_out/r05/call01_pi0.txt
for p in n o p q; do echo "== COVER_TILE_PICK=$p pi0 M=2232"; SHAPES=pi0 COVER_TILE_PICK=$p timeout 300 python tools/dbg/bench_prefill.py 2232 3; done 2>&1 | tee -a gpurun_out/r05/call01_pi0.txt
