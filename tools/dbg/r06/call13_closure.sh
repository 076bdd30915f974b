# round 6, GPU call 13: timeline of the pi0 expert's GEMMs; the round-end measurement pass on the final library
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
EXPERT=1 COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dbg.so timeout 600 python tools/dbg/tiled_timeline.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/c13_expert_gemm_timeline.txt
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dbg.so timeout 600 python tools/decode_budget.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06/c13_decode_layer_budget.txt
bash tools/closure.sh r06
