# round-4 GPU call 2: the persistent decode chain -- parity tests first (bounded), then same-box A/B of the headline with and without it
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_chain_gpu.py -q -x -s 2>&1 | tail -30 > gpurun_out/r04b_chain_tests.log
tail -25 gpurun_out/r04b_chain_tests.log
timeout 600 python -m pytest tests/test_openvla_gpu.py -q -x -s -k "g3" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_fullsize_gpu.py -q -x -k "decision_properties or config2 or config3" 2>&1 | tail -5
for m in 1 0 1 0; do COVER_DECODE_CHAIN=$m python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-profile 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CHAIN=$m', d['ms_per_step'], d['value'])"; done
COVER_DECODE_CHAIN=1 python tools/phases.py 2>/dev/null | tail -1
COVER_DECODE_CHAIN=0 python tools/phases.py 2>/dev/null | tail -1
COVER_DECODE_CHAIN=2 python tools/phases.py 2>/dev/null | tail -1
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r04b_bench_line.json 2>/dev/null; cut -c1-1800 gpurun_out/r04b_bench_line.json
