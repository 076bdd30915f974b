set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "decode_attention" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_openvla_gpu.py tests/test_chain_gpu.py -q -x 2>&1 | tail -3
for i in 1 2 3; do timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('headline', d['ms_per_step'])"; done
python tools/phases.py 2>/dev/null | tail -1
