cd $GRAFT_REPO_ROOT
timeout 900 python tools/dbg/r05/bench_expert.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_bench_expert.txt
