cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 0 1 0 1; do COVER_SIDE_AFTER_VISION=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('SIDE_AFTER_VISION=$m', d['ms_per_step'], d['value'])"; done
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8
python bench.py --dtype fp8 --no-cpu-baseline > gpurun_out/r04d_fp8_n32_bench_line.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04d_fp8_n32_bench_line.json')); print(d['ms_per_step']); print(json.dumps(d.get('fp8_vs_bf16'), indent=0)[:2500])"
