# round 6, GPU call 26: per-block timeline of the config-5 decode attention (key-split kernel resumed from the own-token state)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
SHAPE=c5 COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_atdbg.so timeout 300 python tools/dbg/at_timeline.py 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/c26_attn_timeline.txt
