cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv" > gpurun_out/r2_i_tests.log 2>&1; tail -2 gpurun_out/r2_i_tests.log
python scripts/kbench.py pg > gpurun_out/r2_i_kbench.txt 2>&1; cat gpurun_out/r2_i_kbench.txt
python scripts/kbench.py 5x5 >> gpurun_out/r2_i_kbench.txt 2>&1; tail -1 gpurun_out/r2_i_kbench.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_i_bench.json 2> gpurun_out/r2_i_bench.err; cut -c1-200 gpurun_out/r2_i_bench.json
