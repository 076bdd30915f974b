set -x
cd $GRAFT_REPO_ROOT
export COVER_LIB_PATH=$PWD/build_dbg/libcover_fl.so
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "headline_prefill or long_panel or gemm_random_sweep or gemm_tiled or fused_rmsnorm or fused_layernorm or gemm_epilogues" 2>&1 | tail -5
timeout 300 python tools/dbg/bench_prefill.py 448 3 2>&1 | tail -5
unset COVER_LIB_PATH
timeout 300 python tools/dbg/bench_prefill.py 448 3 2>&1 | tail -5
