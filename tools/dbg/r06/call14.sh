# round 6, GPU call 14: per-block timeline of the fused decode attention (15 us x 192 launches per decision)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
for m in cold back2back; do MODE=$m COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dadbg.so timeout 300 python tools/dbg/exp_da_debug.py 2>&1 | grep -v amdgpu.ids; done | tee $O/c14_da_timeline.txt
