set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_models_gpu.py -q -x 2>&1 | tail -4
for i in 1 2; do
for t in 0 1; do
COVER_ROPE_ATTN_FUSE=$t timeout 600 python bench.py --profile pi0 --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('FUSE=$t', d['ms_per_step'])"
done; done
