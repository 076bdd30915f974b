# round 6, GPU call 25: o_proj of a config-5 decode pass (MX input, + post norm) under other tile / split plans
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do
python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
COVER_TILE_PICK=c COVER_TILE_SPLIT=4 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
COVER_TILE_PICK=c COVER_TILE_SPLIT=2 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
COVER_TILE_PICK=d COVER_TILE_SPLIT=4 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
COVER_TILE_PICK=i COVER_TILE_SPLIT=2 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
COVER_TILE_PICK=i COVER_TILE_SPLIT=3 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
COVER_TILE_PICK=a COVER_TILE_SPLIT=2 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1
done | tee $O/c25_oproj_plans.txt
K=11008 python tools/dbg/bench_oproj_mx.py 2>&1 | tail -1 | tee -a $O/c25_oproj_plans.txt
