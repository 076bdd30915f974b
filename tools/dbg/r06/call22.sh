# round 6, GPU call 22: the attention MX-output test + the attention / decode kernel tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_fp8_gpu.py -q -x -k "attention or decoder_mx" -s 2>&1 | grep -E "MX |passed|failed|FAILED|Error|error|assert" | cut -c1-300 | tail -20 | tee $O/c22_tests.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or decode" 2>&1 | tail -3 | cut -c1-300 | tee -a $O/c22_tests.txt
