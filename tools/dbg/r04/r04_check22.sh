set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d gpurun_out/r04_lds -o lds -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > gpurun_out/r04_lds_stdout.log 2>&1
python tools/pmc_any.py gpurun_out/r04_lds/lds_results.db gemm_tiled > gpurun_out/r04_pmc_lds.txt 2>&1
python tools/pmc_any.py gpurun_out/r04_lds/lds_results.db attn >> gpurun_out/r04_pmc_lds.txt 2>&1
python tools/pmc_any.py gpurun_out/r04_lds/lds_results.db skinny >> gpurun_out/r04_pmc_lds.txt 2>&1
cat gpurun_out/r04_pmc_lds.txt | head -120
rm -rf gpurun_out/r04_lds/*.db
