set -x
cd $GRAFT_REPO_ROOT
COVER_LIB_PATH=$PWD/build_dbg/libcover_dadbg.so MODE=cold python tools/dbg/exp_da_debug.py 2>&1 | tail -2 | cut -c1-400
run() { timeout 600 python bench.py --steps 15 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'])"; }
for i in 1 2 3; do
run new
COVER_LIB_PATH=$PWD/build_dbg/libcover_daold.so run old
done
