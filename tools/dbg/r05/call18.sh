# round 5, GPU call 18: fp8 tiled GEMMs at the config-5 decode shape (M = 512), loader-wave vs self-loading form, per shape and per tile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for v in 0 1; do echo "== COVER_V3_F8=$v"; FP8=1 COVER_V3_F8=$v timeout 300 python tools/dbg/bench_prefill.py 512 3 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/call18_f8_m512.txt
for pk in c d i; do echo "== COVER_V3_F8=1 COVER_TILE_PICK=$pk"; FP8=1 COVER_TILE_PICK=$pk timeout 300 python tools/dbg/bench_prefill.py 512 3 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r05/call18_f8_m512.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fast_activation or tiled" 2>&1 | tail -4
