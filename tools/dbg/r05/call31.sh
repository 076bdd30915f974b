# round 5, GPU call 31: matrix-pipe counters of ONE config-5 decision (fp8 self-loading kernels) through the committed tool
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/r05_c5mf -o mf -- python3 bench.py --dtype fp8 --samples 64 --horizon 8 --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-agreement > gpurun_out/r05/call31_stdout.log 2>&1
ls -la gpurun_out/r05_c5mf | head -5
python tools/pmc_mfma.py gpurun_out/r05_c5mf/mf_results.db > gpurun_out/r05_config5_pmc_mfma.txt 2>&1
head -16 gpurun_out/r05_config5_pmc_mfma.txt | cut -c1-160
rm -rf gpurun_out/r05_c5mf
