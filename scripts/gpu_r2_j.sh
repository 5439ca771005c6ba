cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_identities_gpu.py -m gpu -x -q -k "conv or identit" > gpurun_out/r2_j_tests.log 2>&1; tail -2 gpurun_out/r2_j_tests.log
python scripts/kbench.py > gpurun_out/r2_j_kbench.txt 2>&1; cat gpurun_out/r2_j_kbench.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_j_bench.json 2> gpurun_out/r2_j_bench.err; cut -c1-200 gpurun_out/r2_j_bench.json
