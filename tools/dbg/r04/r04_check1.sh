# round-4 GPU call 1: the new tests first (fast feedback), then the whole -m gpu suite, phases in both side modes, kernel trace with the
# per-pass table, the P1 line, config 3 at W = 1, MFMA PMC pass. Everything lands in gpurun_out/.
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "g3 or m448 or config3_one or pi0_profile_b40 or error_bound or decode_own or sampler_matches_reference" -s 2>&1 | tail -40 > gpurun_out/r04a_newtests.log
tail -15 gpurun_out/r04a_newtests.log
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r04a_gputests.log
tail -8 gpurun_out/r04a_gputests.log
python tools/phases.py > gpurun_out/r04a_phases.txt 2>/dev/null
SIDE_MODE=0 python tools/phases.py >> gpurun_out/r04a_phases.txt 2>/dev/null
SIDE_MODE=1 python tools/phases.py >> gpurun_out/r04a_phases.txt 2>/dev/null
cat gpurun_out/r04a_phases.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/r04a -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r04a_trace_stdout.log 2>&1
python tools/rocpd_stats.py gpurun_out/r04a/bench_results.db patchify_k > gpurun_out/r04a_bench_kernel_stats.txt 2>&1
python tools/timeline.py gpurun_out/r04a/bench_results.db > gpurun_out/r04a_timeline.txt 2>&1
tail -30 gpurun_out/r04a_timeline.txt
python bench.py --profile pi0 > gpurun_out/r04a_pi0_bench_line.json 2> gpurun_out/r04a_pi0_stderr.log
cut -c1-1500 gpurun_out/r04a_pi0_bench_line.json; tail -3 gpurun_out/r04a_pi0_stderr.log
python bench.py --config 3 --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04a_config3_w1_line.json 2>/dev/null
cut -c1-600 gpurun_out/r04a_config3_w1_line.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/r04a_mf -o mf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > gpurun_out/r04a_mf_stdout.log 2>&1
python tools/pmc_mfma.py gpurun_out/r04a_mf/mf_results.db > gpurun_out/r04a_pmc_mfma.txt 2>&1
head -40 gpurun_out/r04a_pmc_mfma.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04a_bench_line.json 2>/dev/null
cut -c1-400 gpurun_out/r04a_bench_line.json
rm -rf gpurun_out/r04a/*.db gpurun_out/r04a_mf/*.db
