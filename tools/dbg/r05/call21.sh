# round 5, GPU call 21: loader-wave / self-loading agreement test; the 64 x 64 tile of ViT-sized GEMMs on the self-loading kernel (COVER_V3_SMALL=1) vs gemm_tiled
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "forms_agree or fast_activation" 2>&1 | tail -4
COVER_V3_SMALL=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -4
for i in 1 2; do for v in 0 1; do echo "== COVER_V3_SMALL=$v headline"; COVER_V3_SMALL=$v timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done; done | tee gpurun_out/r05/call21_small.txt
for v in 0 1; do echo "== COVER_V3_SMALL=$v phases"; COVER_V3_SMALL=$v timeout 600 python tools/phases.py 2>/dev/null | tail -1; done | tee -a gpurun_out/r05/call21_small.txt
for v in 0 1 0 1; do echo "== COVER_V3_SMALL=$v P1"; COVER_V3_SMALL=$v timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee -a gpurun_out/r05/call21_small.txt
