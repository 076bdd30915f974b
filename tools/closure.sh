set -x
SHA=$(sha256sum cover_vla_amd/libcover_hip.so | cut -c1-16)
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r02f -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02f_trace_stdout.log 2>&1
python tools/rocpd_stats.py gpurun_out/r02f/bench_results.db patchify_k > gpurun_out/r02f_bench_kernel_stats.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/r02f_pmc -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > gpurun_out/r02f_pmc_stdout.log 2>&1
python tools/pmc_stats.py gpurun_out/r02f_pmc/pmc_results.db profiles/r02_pmc_traffic.json $SHA > gpurun_out/r02f_pmc_fetch_size.txt 2>&1
cp profiles/r02_pmc_traffic.json gpurun_out/r02_pmc_traffic.json
tail -3 gpurun_out/r02f_pmc_fetch_size.txt
python bench.py > gpurun_out/r02f_bench_line.json 2> gpurun_out/r02f_bench_stderr.log
cat gpurun_out/r02f_bench_line.json
rm -rf gpurun_out/r02f_pmc/*.db gpurun_out/r02f/*.db
