# round 6, GPU call 19: MX block scales after the operand-order fix -- probes, fp8 tests, config 5 A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 300 python tools/dbg/mx_probe2.py 2>&1 | grep -v amdgpu.ids | tail -20 | tee $O/c19_mx_probe2.txt
M=512 N=4096 K=11008 timeout 300 python tools/dbg/mx_probe.py 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/c19_mx_probe.txt
timeout 1500 python -m pytest tests/test_fp8_gpu.py -q -s 2>&1 | grep -E "MX GEMM|passed|failed|FAILED|Error" | cut -c1-400 | tee $O/c19_fp8_tests.txt
for rep in 1 2; do for mx in 0 1; do
  echo "== COVER_FP8_MX=$mx config 5 (rep $rep)"; COVER_FP8_MX=$mx timeout 900 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null > $O/c19_c5_mx${mx}_$rep.json; python -c "import sys,json; d=json.load(open('$O/c19_c5_mx${mx}_$rep.json')); print(d['ms_per_step'], d.get('roofline',{}).get('frac'), json.dumps(d.get('fp8_vs_bf16'))[:1500])"
done; done | tee $O/c19_config5_ab.txt
for mx in 0 1; do echo "== COVER_FP8_MX=$mx fp8 N=32"; COVER_FP8_MX=$mx timeout 600 python bench.py --dtype fp8 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee -a $O/c19_config5_ab.txt
