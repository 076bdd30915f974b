# round 6, GPU call 33: per-wave stamps of one workgroup of the fused decode attention (who is the tile phase waiting for?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
MODE=cold COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dadbg.so timeout 300 python tools/dbg/exp_da_debug.py 2>&1 | grep -v amdgpu.ids | tail -9 | cut -c1-900 | tee $O/c33_da_waves.txt
