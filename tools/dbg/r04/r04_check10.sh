cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_kernels_gpu.py -q -x -k "verifier or gemm_f32 or mha" 2>&1 | tail -3
for m in 8 4 8 4; do COVER_F32_UNR=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('F32_UNR=$m', d['ms_per_step'], d['value'])"; done
COVER_F32_UNR=8 python tools/phases.py 2>/dev/null | tail -1
COVER_F32_UNR=4 python tools/phases.py 2>/dev/null | tail -1
