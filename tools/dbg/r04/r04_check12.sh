set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "score_select" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_openvla_gpu.py -q -x 2>&1 | tail -3
BIG=1 python tools/dbg/exp_f32.py
BIG=1 COVER_F32_DIRECT_MAX=1000000 python tools/dbg/exp_f32.py
M=448 python tools/dbg/exp_blas.py
python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('headline', d['ms_per_step'])"
python tools/phases.py 2>/dev/null | tail -1
