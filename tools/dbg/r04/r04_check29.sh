set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "decode_attention" 2>&1 | tail -3
COVER_LIB_PATH=$PWD/build_dbg/libcover_dadbg.so MODE=cold python tools/dbg/exp_da_debug.py 2>&1 | tail -2 | cut -c1-330
run() { timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
run v4
COVER_LIB_PATH=$PWD/build_dbg/libcover_dav3.so run v3
done
