# round 6, GPU call 39-40: permlane-swap reductions (inline asm): ubench + attention tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/pls tools/ubench/permlane_swap.hip 2>/dev/null; /tmp/pls | tail -3 | cut -c1-330 | tee $O/c40_permlane.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or decode" 2>&1 | tail -4 | cut -c1-250 | tee $O/c40_tests.txt
