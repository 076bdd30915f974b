# round 6, GPU call 16: MX block scales for the down_proj input -- kernel-level tests first (new MX tests, then the whole fp8 file), then config 5 A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_fp8_gpu.py -q -x -k "mx or klinear" -s 2>&1 | grep -v amdgpu.ids | tail -40 | cut -c1-400 | tee $O/c16_mx_tests.txt
timeout 1500 python -m pytest tests/test_fp8_gpu.py -q 2>&1 | tail -5 | cut -c1-600 | tee $O/c16_fp8_tests.txt
for rep in 1 2; do for mx in 0 1; do
  echo "== COVER_FP8_MX=$mx config 5 (rep $rep)"; COVER_FP8_MX=$mx timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d.get('roofline',{}).get('frac'), d.get('fp8_vs_bf16'))"
done; done | cut -c1-600 | tee $O/c16_config5_ab.txt
echo "== fused off"; COVER_FP8_MX_FUSE=0 timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" | tee -a $O/c16_config5_ab.txt
