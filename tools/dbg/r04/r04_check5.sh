cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_chain_gpu.py -q -x 2>&1 | tail -2
python tools/dbg/prof_chain.py 2>&1 | tail -3
COVER_LIB_PATH=$PWD/cover_vla_amd/libcover_hip_dcdbg.so MODE=1 python tools/dbg/dc_timeline.py 2>&1 | grep -E "active|seam passed|x window|loop done|k-sums|stores drained|phase start"
for m in 1 0 1 0; do COVER_DECODE_CHAIN=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CHAIN=$m', d['ms_per_step'], d['value'])"; done
