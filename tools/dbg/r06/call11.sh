# round 6, GPU call 11: splitk_reduce_norm with the bf16 residual chunk requested with the first slab batch (default now) vs behind the slab sums (-DCOVER_RN_RES_EARLY=0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_rn1dbg.so timeout 300 python tools/dbg/rn_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/c11_rn_timeline.txt
for rep in 1 2 3 4; do for v in rn0 new; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = new ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v headline (rep $rep)"; COVER_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c11_rn_early_ab.txt
for rep in 1 2; do for v in rn0 new; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = new ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v P1 (rep $rep)"; COVER_LIB_PATH=$lib timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee -a $O/c11_rn_early_ab.txt
for v in rn0 new; do lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = new ] && lib=$PWD/cover_vla_amd/libcover_hip.so; COVER_LIB_PATH=$lib python tools/phases.py 2>/dev/null | tail -1; done | tee -a $O/c11_rn_early_ab.txt
