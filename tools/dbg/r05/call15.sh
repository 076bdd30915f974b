cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== M=448 probe with the epilogue split"; timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call15_m448.txt
echo "== pi0 M=2232"; SHAPES=pi0 timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call15_pi0.txt
for s in 2.0 2.5; do
  echo "== COVER_SYNTH_SIGMA=$s"
  COVER_SYNTH_SIGMA=$s timeout 900 python bench.py --dtype fp8 --no-cpu-baseline --no-profile 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); f = d['fp8_vs_bf16']['teacher_forced_per_step']
for k, v in f.items(): print(k, [(x['decided_rows'], x['top1_agreement'], x['logit_rel_l2']) for x in v])
print('free-running agreement', d['fp8_vs_bf16']['token_agreement_free_running'], 'winner_same', d['fp8_vs_bf16']['winner_same'])"
done
