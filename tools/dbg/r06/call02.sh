# round 6, GPU call 2: the pruned library (no decode chain / tail / head reduction / deferred norm / fused-RoPE attention / bf16 loader-wave long-panel tiles)
# through the whole GPU suite; the k-split wave-pair kernel (gemm_tiled_v3k) vs the one-wave-per-SIMD 224x96 tile; bench line; pi0 read-span repeat
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "k_split or headline_prefill or long_panel" -s 2>&1 | grep -v amdgpu.ids | tail -15 | tee $O/c02_pair_tests.txt
for rep in 1 2 3; do
for pk in auto r u; do
  echo "== qkv M=448 pick $pk (rep $rep)"; P=""; [ $pk != auto ] && P="COVER_TILE_PICK=$pk"
  env $P SHAPE=qkv timeout 300 python tools/dbg/bench_prefill.py 448 6 2>&1 | grep -v amdgpu.ids | grep qkv
done; done | tee $O/c02_pair_qkv.txt
for pk in auto o v; do
  echo "== o_proj/down M=448 pick $pk"; P=""; [ $pk != auto ] && P="COVER_TILE_PICK=$pk COVER_TILE_SPLIT=4"
  env $P SHAPE=o_proj,down timeout 300 python tools/dbg/bench_prefill.py 448 6 2>&1 | grep -v amdgpu.ids | grep -E "o_proj|down"
done | tee $O/c02_pair_oproj.txt
echo "== layer M=448 default"; timeout 300 python tools/dbg/bench_prefill.py 448 4 2>&1 | grep -v amdgpu.ids | tee $O/c02_layer_m448.txt
echo "== layer pi0 default"; SHAPES=pi0 timeout 300 python tools/dbg/bench_prefill.py 2232 3 2>&1 | grep -v amdgpu.ids | tee $O/c02_layer_pi0.txt
python bench.py --no-cpu-baseline > $O/c02_bench_line.json 2> $O/c02_bench_stderr.log; cut -c1-300 $O/c02_bench_line.json
python bench.py --profile pi0 --no-cpu-baseline > $O/c02_pi0_line.json 2>/dev/null; cut -c1-200 $O/c02_pi0_line.json
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/c02_gputests.txt
