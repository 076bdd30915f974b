# round 6, GPU calls 30-31: segment-0 tiles of the fused decode attention by LDS-DMA (tools/ab/libcover_hip_dadma.so = decode_attn.hip -DCOVER_DA_DMA=1) against the product library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dadma.so timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_openvla_gpu.py -q -x -k "decode or openvla or attention" 2>&1 | tail -4 | cut -c1-300 | tee $O/c31_tests.txt
for lib in dadmadbg; do echo "== $lib"; MODE=cold COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_$lib.so timeout 300 python tools/dbg/exp_da_debug.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-330; done | tee $O/c31_da_timeline.txt
for rep in 1 2 3; do
  echo "== product headline (rep $rep)"; timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== dma headline (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dadma.so timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done | tee $O/c31_headline_ab.txt
