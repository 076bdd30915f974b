# round 6, GPU call 4: suite on the current tree; verifier graph launched from the second thread vs eager; read-span repeat; decode budget; residual-only epilogue
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "k_split or headline_prefill or long_panel or gemm_epilogues or random_sweep" 2>&1 | grep -v amdgpu.ids | tail -6 | tee $O/c04_gemm_tests.txt
for rep in 1 2 3; do for sg in 0 2; do
  echo "== headline side graph=$sg (rep $rep)"; COVER_SIDE_GRAPH=$sg timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c04_side_graph.txt
for rep in 1 2; do for sg in 0 2; do
  echo "== P1 side graph=$sg (rep $rep)"; COVER_SIDE_GRAPH=$sg timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee -a $O/c04_side_graph.txt
for m in 0 6; do SIDE_MODE=$([ $m = 0 ] && echo 4 || echo 6) python tools/phases.py 2>/dev/null | tail -1; done | tee $O/c04_phases.txt
for rep in 1 2 3; do for v in base rd10 rd12; do
  lib=$PWD/tools/ab/libcover_hip_$v.so; [ $v = base ] && lib=$PWD/cover_vla_amd/libcover_hip.so
  echo "== $v pi0 M=2232 (rep $rep)"; COVER_LIB_PATH=$lib SHAPES=pi0 timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -v amdgpu.ids | cut -c1-260
  echo "== $v M=448 (rep $rep)"; COVER_LIB_PATH=$lib timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids | grep layer
done; done | tee $O/c04_rdspan.txt
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dbg.so timeout 600 python tools/decode_budget.py 2>&1 | grep -v amdgpu.ids | tee $O/c04_decode_layer_budget.txt
python bench.py --no-cpu-baseline > $O/c04_bench_line.json 2> $O/c04_bench_stderr.log; cut -c1-200 $O/c04_bench_line.json
python bench.py --profile pi0 --no-cpu-baseline > $O/c04_pi0_line.json 2>/dev/null; cut -c1-200 $O/c04_pi0_line.json
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/c04_gputests.txt
