# round 6, GPU calls 43, 45: per-block timeline of attn_shared_k at the config-5 decode shape (-DCOVER_AT_DEBUG build)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
SHAPE=c5 COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_atdbg.so timeout 300 python tools/dbg/at_timeline.py 2>&1 | grep -v amdgpu.ids | tail -4 | cut -c1-420 | tee $O/c45_attn_shared_timeline.txt
