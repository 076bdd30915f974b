set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_chain_gpu.py -q -x -k tail 2>&1 | tail -5
for i in 1 2; do
for t in 0 1; do
COVER_TAIL_REDUCE=$t timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('TAIL=$t', d['ms_per_step'])"
done; done
COVER_TAIL_REDUCE=1 timeout 600 python tools/phases.py 2>/dev/null | tail -1
