cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests/test_vgg_gpu.py -m gpu -x -q -k "oracle_masks" > gpurun_out/r2_h_tests.log 2>&1; tail -30 gpurun_out/r2_h_tests.log; grep "oracle's saved" gpurun_out/parity_report.txt | tail -4
