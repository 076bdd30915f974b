# round 6, GPU calls 24, 42: kernel table of one config-5 decision (N = 512, horizon 8, fp8) on the final library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/c5trace -o c5 -- python3 bench.py --dtype fp8 --samples 64 --horizon 8 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $O/c42_stdout.log 2>&1
python tools/rocpd_stats.py $O/c5trace/c5_results.db patchify_k > $O/c42_config5_kernel_stats.txt 2>&1
head -40 $O/c42_config5_kernel_stats.txt | cut -c1-260
rm -rf $O/c5trace
