cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_model_gpu.py tests/test_graph_gpu.py -m gpu -x -q -k "sparse or train_gan3_default" > gpurun_out/r2_g_tests.log 2>&1; tail -30 gpurun_out/r2_g_tests.log; grep "sparse_512\|train_gan3 default" gpurun_out/parity_report.txt | tail -4
