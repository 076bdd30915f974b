# round 5, GPU call 23: pi0 prefix o_proj (M = 2232, N = 2048, K = 2048): unsplit tiles + a norm launch against the planner's split 224 x 192 + reduction
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== default"; SHAPES=pi0 SHAPE=o_proj timeout 120 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep "^M=" | tee gpurun_out/r05/call23_pi0_oproj.txt
for pk in r o n p q; do for sp in 1 2; do echo "== pick $pk split $sp"; SHAPES=pi0 SHAPE=o_proj COVER_TILE_PICK=$pk COVER_TILE_SPLIT=$sp timeout 120 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep "^M="; done; done | tee -a gpurun_out/r05/call23_pi0_oproj.txt
