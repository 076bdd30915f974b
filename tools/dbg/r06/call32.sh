# round 6, GPU call 32: the tail tile of the fused decode attention by LDS-DMA (product library) against tools/ab/libcover_hip_danodma.so (decode_attn.hip -DCOVER_DA_DMA=0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_openvla_gpu.py -q -x -k "decode or openvla or attention" 2>&1 | tail -3 | cut -c1-300 | tee $O/c32_tests.txt
for lib in dadbg dadmadbg; do echo "== $lib"; MODE=cold COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_$lib.so timeout 300 python tools/dbg/exp_da_debug.py 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-330; done | tee $O/c32_da_timeline.txt
for rep in 1 2 3; do
  echo "== no dma headline (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_danodma.so timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== tail dma headline (rep $rep)"; timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done | tee $O/c32_headline_ab.txt
