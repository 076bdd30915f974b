# round 6, GPU call 9: residual chunk of splitk_reduce_norm requested with the slabs (-DCOVER_RN_RES_EARLY=1) vs behind them (default): timeline + decision A/B
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_rnearlydbg.so timeout 300 python tools/dbg/rn_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/c09_rn_timeline_early.txt
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_rnearly.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "fused_rmsnorm or headline_prefill or gemm_skinny" 2>&1 | tail -2
for rep in 1 2 3; do for v in base rnearly; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = base ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v headline (rep $rep)"; COVER_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c09_rn_early_ab.txt
