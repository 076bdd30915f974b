# round 6, GPU call 6: the 224 x 64 pair tile (pick v) with two K slices against 224 x 128 with four, o_proj / down at M = 448 and the pi0 prefix shapes
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COVER_TILE_PICK=v COVER_TILE_SPLIT=2 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "headline_prefill and (4096-4096 or 4096-11008)" 2>&1 | tail -3 | tee $O/c06_tests.txt
for rep in 1 2 3; do
  echo "== auto (rep $rep)"; SHAPE=o_proj,down timeout 300 python tools/dbg/bench_prefill.py 448 6 2>&1 | grep -E "o_proj|down" | cut -c1-250
  for sp in 2 4; do echo "== pick v split $sp (rep $rep)"; COVER_TILE_PICK=v COVER_TILE_SPLIT=$sp SHAPE=o_proj,down timeout 300 python tools/dbg/bench_prefill.py 448 6 2>&1 | grep -E "o_proj|down" | cut -c1-250; done
done | tee $O/c06_pair64.txt
for pk in auto v; do for sp in 1 2; do
  [ $pk = auto ] && [ $sp = 2 ] && continue
  echo "== pi0 o_proj/down pick $pk split $sp"; P=""; [ $pk != auto ] && P="COVER_TILE_PICK=$pk COVER_TILE_SPLIT=$sp"
  env $P SHAPES=pi0 SHAPE=o_proj,down timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -E "o_proj|down" | cut -c1-250
done; done | tee -a $O/c06_pair64.txt
