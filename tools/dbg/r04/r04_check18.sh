set -x
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for t in 0 1; do
COVER_SIDE_LATE_SUBMIT=$t timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('LATE=$t', d['ms_per_step'])"
done; done
