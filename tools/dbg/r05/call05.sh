# round 5, GPU call 5: v3 picks in the cost model (qkv on 224x96 one-wave-per-SIMD tiles), peaked checkpoint, headline bench, M = 704 tiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
echo "== M=448 auto"; timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call05_m448.txt
for v in 0 1; do echo "== M=704 COVER_V3_BIG=$v"; COVER_V3_BIG=$v timeout 300 python tools/dbg/bench_prefill.py 704 3; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/call05_m704.txt
echo "== M=704 COVER_V3=0"; COVER_V3=0 timeout 300 python tools/dbg/bench_prefill.py 704 3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r05/call05_m704.txt
timeout 1200 python -m pytest tests/test_openvla_gpu.py -x -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -x -q -s -k "decision_properties or config2" 2>&1 | grep -E "greedy M|passed|failed|Error|assert" | head
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r05/call05_bench_line.json 2> gpurun_out/r05/call05_bench_stderr.log; cut -c1-1500 gpurun_out/r05/call05_bench_line.json
COVER_V3=0 timeout 900 python bench.py --no-cpu-baseline --no-profile --steps 10 | cut -c1-300
timeout 900 python bench.py --no-cpu-baseline --no-profile --flat-weights --steps 10 | cut -c1-300
