set -x
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
run base
COVER_ATTN_KSPLIT_MAX=600 run ksplit600
COVER_ATTN_KSPLIT_MAX=520 run ksplit520
done
