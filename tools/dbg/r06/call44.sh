# round 6, GPU call 44: attn_shared_k with two 32-key tiles per softmax step (tools/ab/libcover_hip_attnpair.so = attention.hip -DCOVER_ATTN_PAIR_TILES=1) against the product library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_attnpair.so timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fp8_gpu.py -q -k "attention or decoder_mx" 2>&1 | tail -8 | cut -c1-300 | tee $O/c44_tests.txt
for lib in product attnpair; do echo "== $lib"; if [ $lib = product ]; then SHAPE=c5 timeout 300 python tools/dbg/at_timeline.py 2>&1 | grep "config-5" | cut -c1-100; else SHAPE=c5 COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_$lib.so timeout 300 python tools/dbg/at_timeline.py 2>&1 | grep "config-5" | cut -c1-100; fi; done | tee $O/c44_ab.txt
for rep in 1 2 3; do
  echo "== product config 5 (rep $rep)"; timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== pair config 5 (rep $rep)"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_attnpair.so timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done | tee -a $O/c44_ab.txt
