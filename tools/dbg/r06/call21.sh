# round 6, GPU call 21: MX block scales for the o_proj input as well (attention kernel writes them; 64 x 128 loader-wave tile reads them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_fp8_gpu.py -q -x -s 2>&1 | grep -E "MX |passed|failed|FAILED|Error|error|assert" | cut -c1-300 | tail -30 | tee $O/c21_fp8_tests.txt
timeout 1500 python -m pytest tests/test_fullsize_gpu.py -q -x -k "config5" 2>&1 | tail -15 | cut -c1-300 | tee $O/c21_fullsize_c5.txt
for rep in 1 2; do for mx in 0 down 1; do
  echo "== COVER_FP8_MX=$mx config 5 (rep $rep)"; COVER_FP8_MX=$mx timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > $O/c21_c5_mx${mx}_$rep.json; python -c "import sys,json; d=json.load(open('$O/c21_c5_mx${mx}_$rep.json')); f=d['fp8_vs_bf16']; t=f['teacher_forced_per_step']['weights_and_activations_e4m3 (as run)']; print(d['ms_per_step'], d.get('roofline',{}).get('frac'), f['first_token_agreement'], f['score_rmse'], [(s['logit_rel_l2'], s['decided_rows'], s['decided_agreement']) for s in t])"
done; done | tee $O/c21_config5_ab.txt
echo "== fused off"; COVER_FP8_MX_FUSE=0 timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" | tee -a $O/c21_config5_ab.txt
