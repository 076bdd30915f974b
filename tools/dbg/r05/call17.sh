# round 5, GPU call 17: cpr == 1 fix of the staged epilogue's reciprocal indexing; the self-loading fp8 kernel (gemm_tiled_v3_f8) vs the loader-wave form
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -4
timeout 600 python tools/dbg/fuzz_gemm.py 300 7 2>&1 | tail -8
timeout 900 python -m pytest tests/test_fp8_gpu.py -x -q 2>&1 | tail -6
for v in 0 1; do echo "== COVER_V3_F8=$v fp8 N=32"; COVER_V3_F8=$v timeout 600 python bench.py --dtype fp8 --no-cpu-baseline --no-profile --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee gpurun_out/r05/call17_fp8.txt
for v in 0 1 0 1; do echo "== COVER_V3_F8=$v config 5"; COVER_V3_F8=$v timeout 600 python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; done | tee -a gpurun_out/r05/call17_fp8.txt
for v in 0 1; do echo "== COVER_V3_F8=$v kernel stats"; COVER_V3_F8=$v timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r05/c17_$v -o c5 -- python3 bench.py --dtype fp8 --samples 64 --horizon 8 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1; f=$(ls gpurun_out/r05/c17_$v/*/c5_kernel_stats.csv gpurun_out/r05/c17_$v/c5_kernel_stats.csv 2>/dev/null | head -1); head -12 $f | cut -c1-170; find gpurun_out/r05/c17_$v -name "*.db" -delete; find gpurun_out/r05/c17_$v -name "*trace.csv" -delete; done | tee gpurun_out/r05/call17_c5_stats.txt
