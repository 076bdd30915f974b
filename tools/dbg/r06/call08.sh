# round 6, GPU call 8: per-block timeline of splitk_reduce_norm at the decode shape (2 x 5.5 us per layer-step = 2.1 ms per decision)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_rndbg.so timeout 300 python tools/dbg/rn_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/c08_rn_timeline.txt
