set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_chain_gpu.py -q -x -s 2>&1 | tail -12
timeout 300 python tools/dbg/prof_chain.py 2>&1 | tail -5
rocprofv3 --kernel-trace -d gpurun_out/pc -o pc -- python3 tools/dbg/prof_chain.py > /dev/null 2>&1
python tools/dbg/prof_chain.py --parse gpurun_out/pc/pc_results.db 2>&1 | head -75 > gpurun_out/r04c_chain_phases.txt; head -72 gpurun_out/r04c_chain_phases.txt
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "decision_properties" 2>&1 | grep -v "^$" | tail -45
for m in 1 0; do COVER_DECODE_CHAIN=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CHAIN=$m', d['ms_per_step'], d['value'])"; done
rm -rf gpurun_out/pc/*.db
