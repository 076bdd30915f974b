cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/r05/expert -o ex -- python3 tools/dbg/r05/bench_expert.py > /dev/null 2>&1
python tools/rocpd_stats.py gpurun_out/r05/expert/ex_results.db > gpurun_out/r05_bench_expert_kernel_stats.txt 2>&1
head -24 gpurun_out/r05_bench_expert_kernel_stats.txt
rm -rf gpurun_out/r05/expert
