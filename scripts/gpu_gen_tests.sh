# new generator topologies + their kernels + the graph tests that share the executor
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_generators_gpu.py tests/test_graph_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "generators or graph or resize or crop or dropout or unetish or skip_con" > gpurun_out/gen_tests.log 2>&1
rc=$?; echo "exit=$rc"; tail -40 gpurun_out/gen_tests.log | cut -c1-400
