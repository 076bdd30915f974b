# round 5, GPU call 8: deferred RMSNorm (pi0 expert: five launches per layer-step) -- kernel test, pi0 goldens, P1 A/B; then the whole -m gpu suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "deferred" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_models_gpu.py -x -q -k "pi0" 2>&1 | tail -8
for v in 0 1 0 1; do echo "== COVER_DEFER_NORM=$v"; COVER_DEFER_NORM=$v timeout 600 python bench.py --profile pi0 --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee gpurun_out/r05/call08_gputests.txt
