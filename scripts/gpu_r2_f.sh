cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py -m gpu -x -q > gpurun_out/r2_f_tests.log 2>&1; tail -30 gpurun_out/r2_f_tests.log; grep "make_upscaler_attention" gpurun_out/parity_report.txt | tail -2
