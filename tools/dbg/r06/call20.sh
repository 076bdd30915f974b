# round 6, GPU call 20: MX block scales -- fp8 test file, kernel / model / openvla tests, the full-size config-5 and fp8 tests
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_fp8_gpu.py -q -s 2>&1 | grep -E "MX |rest of the product|passed|failed|FAILED|Error|error" | cut -c1-400 | tee $O/c20_fp8_tests.txt
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_openvla_gpu.py tests/test_fullsize_gpu.py -q 2>&1 | tail -5 | cut -c1-600 | tee $O/c20_tests.txt
