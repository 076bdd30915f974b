# round 5, GPU call 19: fp8 o_proj / down at M = 512 (config-5 decode rows): tile x K-slice sweep (the reduction + norm + e4m3 twin launch is inside the time)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for pk in a c d f i; do for sp in 1 2 3 4; do echo "== pick $pk split $sp"; FP8=1 SHAPE=o_proj,down COVER_TILE_PICK=$pk COVER_TILE_SPLIT=$sp timeout 120 python tools/dbg/bench_prefill.py 512 3 2>&1 | grep "^M="; done; done | tee gpurun_out/r05/call19_f8_sweep.txt
echo "== default"; FP8=1 SHAPE=o_proj,down timeout 120 python tools/dbg/bench_prefill.py 512 3 2>&1 | grep "^M=" | tee -a gpurun_out/r05/call19_f8_sweep.txt
