# round 5, GPU call 24: four operand chunks per thread in flight in the generic store loop of the one-wave-per-SIMD kernels (pi0 prefix o_proj: 224 x 96 with a residual)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
for lib in tools/ab/libcover_hip_r05b.so cover_vla_amd/libcover_hip.so; do echo "== $lib pi0 M=2232"; COVER_LIB_PATH=$PWD/$lib SHAPES=pi0 timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep "^M=\|^layer"; done | tee gpurun_out/r05/call24_pi0.txt
for lib in tools/ab/libcover_hip_r05b.so cover_vla_amd/libcover_hip.so tools/ab/libcover_hip_r05b.so cover_vla_amd/libcover_hip.so; do echo "== $lib P1"; COVER_LIB_PATH=$PWD/$lib timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee -a gpurun_out/r05/call24_pi0.txt
timeout 900 python -m pytest tests/test_models_gpu.py -x -q 2>&1 | tail -3
