# round 6, GPU call 12: + bias / residual early for the towers' reductions (new) vs residual-only early (resonly) vs neither (rn0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_openvla_gpu.py -q 2>&1 | tail -3 | tee $O/c12_tests.txt
for rep in 1 2 3 4; do for v in resonly new; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = new ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v headline (rep $rep)"; COVER_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c12_bias_early_ab.txt
for rep in 1 2 3; do for v in resonly new; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = new ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v P1 (rep $rep)"; COVER_LIB_PATH=$lib timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee -a $O/c12_bias_early_ab.txt
