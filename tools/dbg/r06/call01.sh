# round 6, GPU call 1: full-size oracle agreement tests; LDS / wait counter passes on the self-loading tiled GEMMs;
# A/B of the fragment-read placement / priority / de-phasing variants of gemm_v3.hip (item 1b)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -k oracle_agreement -s 2>&1 | grep -v amdgpu.ids | tail -12 | tee $O/c01_agree.txt
LDS="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
WT="SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"
run_pmc() { # tag, counters, then the bench_prefill arguments (environment already set by the caller)
  tag=$1; ctr=$2; shift 2
  timeout 600 rocprofv3 --pmc $ctr --kernel-trace -d $O/pmc_$tag -o p -- python3 tools/dbg/bench_prefill.py "$@" > $O/pmc_${tag}_stdout.log 2>&1
  python tools/pmc_any.py $O/pmc_$tag/p_results.db gemm_tiled > $O/c01_pmc_$tag.txt 2>&1
  rm -rf $O/pmc_$tag
}
run_pmc lds_m448 "$LDS" 448 1
run_pmc wait_m448 "$WT" 448 1
SHAPES=pi0 run_pmc lds_pi0 "$LDS" 2232 1
FP8=1 run_pmc lds_f8 "$LDS" 512 1
head -50 $O/c01_pmc_lds_m448.txt
# correctness of every variant on the headline prefill tiles, then the A/B
for v in rd12 rd10 rd16 prio1 prio2 dph rd12dph rd12prio1; do
  echo "== $v"; COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_$v.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "headline_prefill or long_panel" 2>&1 | tail -2
done 2>&1 | tee $O/c01_variant_tests.txt
for rep in 1 2; do
for v in base rd12 rd10 rd16 prio1 prio2 dph rd12dph rd12prio1; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = base ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v M=448 (rep $rep)"; COVER_LIB_PATH=$lib timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids
done; done | tee $O/c01_ab_m448.txt
for v in base rd12 rd10 rd16 prio1 prio2 dph rd12dph rd12prio1; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = base ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v pi0 M=2232"; COVER_LIB_PATH=$lib SHAPES=pi0 timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -v amdgpu.ids
done | tee $O/c01_ab_pi0.txt
python bench.py --no-cpu-baseline > $O/c01_bench_line.json 2> $O/c01_bench_stderr.log; cut -c1-600 $O/c01_bench_line.json
