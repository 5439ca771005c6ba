cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "generic_conv" > gpurun_out/r2_c_tests.log 2>&1; tail -25 gpurun_out/r2_c_tests.log; grep "generic bf16" gpurun_out/parity_report.txt | tail -12
