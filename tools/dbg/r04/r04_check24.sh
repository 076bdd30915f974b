set -x
cd $GRAFT_REPO_ROOT
echo "== base"; python tools/dbg/bench_prefill.py 448 3 2>&1 | tail -5
for a in 12 4 1; do echo "== ABL $a"; COVER_LIB_PATH=$PWD/build_dbg/libcover_abl$a.so python tools/dbg/bench_prefill.py 448 3 2>&1 | tail -5; done
