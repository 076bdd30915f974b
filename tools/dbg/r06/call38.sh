# round 6, GPU call 38: what the permlane swap builtins return
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pls tools/ubench/permlane_swap.hip 2>&1 | tail -3; /tmp/pls | tee $O/c38_permlane.txt
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention_len_mask" 2>&1 | tail -25 | cut -c1-200
