# round 5, GPU call 25: ViT-sized GEMMs on the 64 x 64 self-loading tile with a DEEP ring (6 / 8 stages: one block per CU, the whole short K range in flight)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
COVER_V3_SMALL=1 COVER_V3_RING=88 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -3
for cfg in "0 0" "1 0" "1 66" "1 88" "0 0" "1 88"; do set -- $cfg; echo "== COVER_V3_SMALL=$1 COVER_V3_RING=$2 headline"; COVER_V3_SMALL=$1 COVER_V3_RING=$2 timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee gpurun_out/r05/call25_ring.txt
for cfg in "0 0" "1 88"; do set -- $cfg; echo "== COVER_V3_SMALL=$1 COVER_V3_RING=$2 phases"; COVER_V3_SMALL=$1 COVER_V3_RING=$2 timeout 600 python tools/phases.py 2>/dev/null | tail -1; done | tee -a gpurun_out/r05/call25_ring.txt
for cfg in "0 0" "1 88" "0 0" "1 88"; do set -- $cfg; echo "== COVER_V3_SMALL=$1 COVER_V3_RING=$2 P1"; COVER_V3_SMALL=$1 COVER_V3_RING=$2 timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee -a gpurun_out/r05/call25_ring.txt
