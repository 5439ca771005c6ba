cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "norm" > gpurun_out/r2_e_tests.log 2>&1; tail -3 gpurun_out/r2_e_tests.log
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_e_bench_bf16.json 2> gpurun_out/r2_e_bench.err; cut -c1-330 gpurun_out/r2_e_bench_bf16.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2_e_bench_c4.json 2>> gpurun_out/r2_e_bench.err; cut -c1-330 gpurun_out/r2_e_bench_c4.json
