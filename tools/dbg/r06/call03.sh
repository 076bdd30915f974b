# round 6, GPU call 3: whole GPU suite on the pruned tree with the pi0 row-group chains / verifier shared-embeddings graph; A/B of both; decode layer
# budget and small-GEMM timeline from the debug-stamped build
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_openvla_gpu.py -q -k "k_split or pi0_sampler or shared_embeddings" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/c03_new_tests.txt
for rep in 1 2; do for cfg in "1 0" "1 1" "2 1" "4 1" "8 1"; do set -- $cfg
  echo "== P1 chains=$1 graph=$2 (rep $rep)"; COVER_PI0_CHAINS=$1 COVER_PI0_GRAPH=$2 timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c03_pi0_chains.txt
for rep in 1 2; do for sg in 0 1; do
  echo "== P1 side graph=$sg (chains 2, rep $rep)"; COVER_SIDE_GRAPH=$sg timeout 600 python bench.py --profile pi0 --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
  echo "== headline side graph=$sg (rep $rep)"; COVER_SIDE_GRAPH=$sg timeout 600 python bench.py --no-cpu-baseline --no-profile --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"
done; done | tee $O/c03_side_graph.txt
python tools/phases.py > $O/c03_phases.txt 2>/dev/null; cat $O/c03_phases.txt
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dbg.so timeout 600 python tools/decode_budget.py 2>&1 | grep -v amdgpu.ids | tee $O/c03_decode_layer_budget.txt
COVER_LIB_PATH=$PWD/tools/ab/libcover_hip_dbg.so timeout 600 python tools/dbg/tiled_timeline.py 2>&1 | grep -v amdgpu.ids | tee $O/c03_small_gemm_timeline.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee $O/c03_gputests.txt
