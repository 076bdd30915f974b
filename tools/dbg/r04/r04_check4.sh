set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace -d gpurun_out/pc -o pc -- python3 tools/dbg/prof_chain.py > /dev/null 2>&1
python tools/dbg/prof_chain.py --parse gpurun_out/pc/pc_results.db 2>&1 | head -160 > gpurun_out/r04c_chain_phases.txt; sed -n 60,150p gpurun_out/r04c_chain_phases.txt
python tools/dbg/greedy_margin.py 2>&1 | tail -24
rm -rf gpurun_out/pc/*.db
