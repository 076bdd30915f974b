# round 6, GPU call 34: pool C's slot loads requested with the index loads (product library) against tools/ab/libcover_hip_daold.so (decode_attn.hip of the previous commit)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_openvla_gpu.py -q -x -k "decode or openvla or attention" 2>&1 | tail -3 | cut -c1-300 | tee $O/c34_tests.txt
MODE=cold COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dadbg.so timeout 300 python tools/dbg/exp_da_debug.py 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-900 | tee $O/c34_da_waves.txt
for rep in 1 2 3 4; do
  echo "== old headline (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_daold.so timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== new headline (rep $rep)"; timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done | tee $O/c34_headline_ab.txt
