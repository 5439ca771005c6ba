# new generator topologies + their kernels + the graph tests that share the executor; then every fp32 conv / wgrad kernel test (the
# weight-gradient kernel was re-templated on KH, KW)
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/test_generators_gpu.py tests/test_graph_gpu.py tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/gen_tests.log 2>&1
rc=$?; echo "exit=$rc"; tail -30 gpurun_out/gen_tests.log | cut -c1-300
