# round 5, GPU call 26: head reduction (the split-K reduction + norm of o_proj / down in the first workgroups of the next streaming launch) -- parity and A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_chain_gpu.py -x -q 2>&1 | tail -4
for i in 1 2; do for v in old 0 1; do echo "== COVER_HEAD_REDUCE=$v headline"; L=$PWD/cover_vla_amd/libcover_hip.so; if [ $v = old ]; then L=$PWD/tools/ab/libcover_hip_r05b.so; fi; COVER_LIB_PATH=$L COVER_HEAD_REDUCE=$v timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done; done | tee gpurun_out/r05/call26_head.txt
for v in 0 1; do echo "== COVER_HEAD_REDUCE=$v phases"; COVER_HEAD_REDUCE=$v timeout 600 python tools/phases.py 2>/dev/null | tail -1; done | tee -a gpurun_out/r05/call26_head.txt
for v in 0 1; do echo "== COVER_HEAD_REDUCE=$v fp8 N=32"; COVER_HEAD_REDUCE=$v timeout 600 python bench.py --dtype fp8 --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee -a gpurun_out/r05/call26_head.txt

