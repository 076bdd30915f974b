# round 5, GPU call 16: epilogue work (fast bf16 activations, GLU loop with the activation hoisted, reciprocal-multiply indexing, two-pass staging)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -4
for lib in tools/ab/libcover_hip_pre_epi.so cover_vla_amd/libcover_hip.so; do echo "== $lib M=448"; COVER_LIB_PATH=$PWD/$lib timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/call16_m448.txt
for lib in tools/ab/libcover_hip_pre_epi.so cover_vla_amd/libcover_hip.so; do echo "== $lib pi0 M=2232"; COVER_LIB_PATH=$PWD/$lib SHAPES=pi0 timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r05/call16_pi0.txt
for i in 1 2; do for lib in tools/ab/libcover_hip_pre_epi.so cover_vla_amd/libcover_hip.so; do echo "== $lib headline"; COVER_LIB_PATH=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done; done
for lib in tools/ab/libcover_hip_pre_epi.so cover_vla_amd/libcover_hip.so; do echo "== $lib P1"; COVER_LIB_PATH=$PWD/$lib timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r05/call16_gputests.txt
