# round 6, GPU call 15: five-stage ring for the 64 x 64 tile on grids of at most two workgroups per CU (COVER_TILED_DEEP=0 = three stages)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_openvla_gpu.py -q 2>&1 | tail -3 | tee $O/c15_tests.txt
for d in 0 1; do for e in 0 1; do echo "== COVER_TILED_DEEP=$d EXPERT=$e"; COVER_TILED_DEEP=$d EXPERT=$e COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dbg.so timeout 600 python tools/dbg/tiled_timeline.py 2>&1 | grep -v amdgpu.ids | cut -c1-330; done; done | tee $O/c15_deep_timeline.txt
for rep in 1 2 3; do for d in 0 1; do
  echo "== deep=$d P1 (rep $rep)"; COVER_TILED_DEEP=$d timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== deep=$d headline (rep $rep)"; COVER_TILED_DEEP=$d timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c15_deep_ab.txt
for d in 0 1; do COVER_TILED_DEEP=$d python tools/phases.py 2>/dev/null | tail -1; done | tee -a $O/c15_deep_ab.txt
