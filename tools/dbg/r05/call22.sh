# round 5, GPU call 22: 112 x 128 self-loading tile (pick s) with half the K slices for the narrow prefill outputs (o_proj, down) at M = 448 and M = 2232
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== default"; SHAPE=o_proj,down timeout 120 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep "^M=" | tee gpurun_out/r05/call22_112.txt
for sp in 1 2 3 4; do echo "== pick s (112x128) split $sp"; SHAPE=o_proj,down COVER_TILE_PICK=s COVER_TILE_SPLIT=$sp timeout 120 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep "^M="; done | tee -a gpurun_out/r05/call22_112.txt
for sp in 2 3 4; do echo "== pick o (224x128) split $sp"; SHAPE=o_proj,down COVER_TILE_PICK=o COVER_TILE_SPLIT=$sp timeout 120 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep "^M="; done | tee -a gpurun_out/r05/call22_112.txt
echo "== pi0 default"; SHAPES=pi0 SHAPE=o_proj,qkv timeout 120 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep "^M=" | tee -a gpurun_out/r05/call22_112.txt
for sp in 1 2; do echo "== pi0 pick s split $sp"; SHAPES=pi0 SHAPE=o_proj,qkv COVER_TILE_PICK=s COVER_TILE_SPLIT=$sp timeout 120 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep "^M="; done | tee -a gpurun_out/r05/call22_112.txt
COVER_TILE_PICK=s timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "long_panel or headline" 2>&1 | tail -3
