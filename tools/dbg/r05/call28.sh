# round 5, GPU call 28: kernel table of ONE config-5 decision (N = 512, horizon 8, fp8) on the final build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 rocprofv3 --kernel-trace --stats -d gpurun_out/r05_c5 -o c5 -- python3 bench.py --dtype fp8 --samples 64 --horizon 8 --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-agreement > gpurun_out/r05/call28_stdout.log 2>&1
ls -la gpurun_out/r05_c5 | head
python tools/rocpd_stats.py gpurun_out/r05_c5/c5_results.db patchify_k > gpurun_out/r05_config5_kernel_stats.txt 2>&1
head -30 gpurun_out/r05_config5_kernel_stats.txt | cut -c1-170
rm -rf gpurun_out/r05_c5
