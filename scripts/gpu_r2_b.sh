cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv" > gpurun_out/r2_b_tests.log 2>&1; tail -3 gpurun_out/r2_b_tests.log
python scripts/kbench.py > gpurun_out/r2_b_kbench.txt 2>&1; cat gpurun_out/r2_b_kbench.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_b_bench.json 2> gpurun_out/r2_b_bench.err; cut -c1-400 gpurun_out/r2_b_bench.json
