set -x
cd $GRAFT_REPO_ROOT
run() { timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], 'prefill', d['roofline_mfma']['prefill'], 'vit', d['roofline_mfma']['vit']['frac'])"; }
for i in 1 2; do
run base
COVER_LIB_PATH=$PWD/build_dbg/libcover_wnt.so run wnt
done
