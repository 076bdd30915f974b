# round 6, GPU calls 27-28: the workgroup-shared-keys attention form (attn_shared_k) -- tests, timeline, config-5 A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or decode" 2>&1 | tail -12 | cut -c1-300 | tee $O/c28_tests.txt
timeout 900 python -m pytest tests/test_fp8_gpu.py -q -x -k "attention or decoder_mx" 2>&1 | tail -5 | cut -c1-300 | tee -a $O/c28_tests.txt
for sh in 1 0; do
  echo "== COVER_ATTN_SHARED=$sh"; SHAPE=c5 COVER_ATTN_SHARED=$sh timeout 300 python tools/dbg/at_timeline.py 2>&1 | grep "config-5" | cut -c1-120
done | tee $O/c28_attn_ab.txt
for rep in 1 2; do for sh in 0 1; do
  echo "== COVER_ATTN_SHARED=$sh config 5 (rep $rep)"; COVER_ATTN_SHARED=$sh timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee -a $O/c28_attn_ab.txt
