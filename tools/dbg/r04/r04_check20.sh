set -x
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for t in 0 1; do
COVER_ROPE_ATTN_FUSE=$t timeout 600 python bench.py --profile pi0 --steps 30 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('FUSE=$t', d['ms_per_step'])"
done; done
