cd $GRAFT_REPO_ROOT
timeout 1500 python tools/dbg/r05/peaked_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05_peaked_sweep.txt
