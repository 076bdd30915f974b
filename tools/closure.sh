# Round-end measurement pass (run through gpurun from the repo root): full -m gpu suite, smoke, kernel trace of NORMAL decisions (+ per-pass
# table), PMC FETCH_SIZE pass of the SAME library build (its sha16 goes into profiles/<tag>_pmc_traffic.json), the bench line with its CPU
# leg, the P1 line with its own kernel table, and the secondary configurations. Everything lands in gpurun_out/ (merged back); copy what is
# to be judged into profiles/.
set -x
TAG=${1:-r06}
SHA=$(sha256sum cover_vla_amd/libcover_hip.so | cut -c1-16)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/${TAG}_gputests.txt; cat gpurun_out/${TAG}_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG} -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_trace_stdout.log 2>&1
python tools/rocpd_stats.py gpurun_out/${TAG}/bench_results.db patchify_k > gpurun_out/${TAG}_bench_kernel_stats.txt 2>&1
python tools/timeline.py gpurun_out/${TAG}/bench_results.db > gpurun_out/${TAG}_timeline.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/${TAG}_pmc -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > gpurun_out/${TAG}_pmc_stdout.log 2>&1
python tools/pmc_stats.py gpurun_out/${TAG}_pmc/pmc_results.db profiles/${TAG}_pmc_traffic.json $SHA > gpurun_out/${TAG}_pmc_fetch_size.txt 2>&1
cp profiles/${TAG}_pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
tail -1 gpurun_out/${TAG}_pmc_fetch_size.txt
# matrix-pipe utilisation (own PMC pass, kernel trace only) through the COMMITTED tool
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/${TAG}_mf -o mf -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2>&1
python tools/pmc_mfma.py gpurun_out/${TAG}_mf/mf_results.db > gpurun_out/${TAG}_pmc_mfma.txt 2>&1; head -12 gpurun_out/${TAG}_pmc_mfma.txt
python bench.py > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_stderr.log
cut -c1-400 gpurun_out/${TAG}_bench_line.json
python tools/phases.py > gpurun_out/${TAG}_phases.txt 2>/dev/null; cat gpurun_out/${TAG}_phases.txt
python bench.py --profile pi0 > gpurun_out/${TAG}_pi0_bench_line.json 2>/dev/null; cut -c1-300 gpurun_out/${TAG}_pi0_bench_line.json
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_pi0 -o pi0 -- python3 bench.py --profile pi0 --steps 3 --warmup 1 --no-cpu-baseline --no-profile > /dev/null 2>&1
python tools/rocpd_stats.py gpurun_out/${TAG}_pi0/pi0_results.db patchify_k > gpurun_out/${TAG}_pi0_kernel_stats.txt 2>&1
python bench.py --config 3 --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_config3_w1_line.json 2>/dev/null
python bench.py --dtype fp8 --no-cpu-baseline > gpurun_out/${TAG}_fp8_n32_bench_line.json 2>/dev/null
python bench.py --dtype fp8 --samples 64 --horizon 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_config5_fp8_n512_h8_bench_line.json 2>/dev/null
python bench.py --samples 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_config2_n16_bench_line.json 2>/dev/null
python bench.py --samples 8 --cams 2 --members 2 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_config4_2cam_n64_bench_line.json 2>/dev/null
python tools/bench_pi0fast.py 2>/dev/null | tail -1 > gpurun_out/${TAG}_pi0fast_line.json
for f in config3_w1_line fp8_n32_bench_line config5_fp8_n512_h8_bench_line config2_n16_bench_line config4_2cam_n64_bench_line pi0_bench_line; do python -c "import json,sys; d=json.load(open('gpurun_out/${TAG}_'+sys.argv[1]+'.json')); print(sys.argv[1], d['ms_per_step'], d['value'])" $f; done
cat gpurun_out/${TAG}_pi0fast_line.json
rm -rf gpurun_out/${TAG}_pmc/*.db gpurun_out/${TAG}/*.db gpurun_out/${TAG}_pi0/*.db gpurun_out/${TAG}_mf/*.db
