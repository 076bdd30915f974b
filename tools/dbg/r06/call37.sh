# round 6, GPU calls 37, 41: cross-row reductions of the attention kernels on v_permlane16_swap / v_permlane32_swap (product library) against tools/ab/libcover_hip_prev.so (ds_bpermute)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_openvla_gpu.py tests/test_fp8_gpu.py -q -x 2>&1 | tail -3 | cut -c1-300 | tee $O/c41_tests.txt
for rep in 1 2 3; do
  echo "== prev headline (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_prev.so timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== new headline (rep $rep)"; timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done | tee $O/c41_ab.txt
for rep in 1 2; do
  echo "== prev config 5 (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_prev.so timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== new config 5 (rep $rep)"; timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== prev P1 (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_prev.so timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== new P1 (rep $rep)"; timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done | tee -a $O/c41_ab.txt
