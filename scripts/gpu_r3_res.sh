# residual-tile request earlier in the v1 trunk kernel; cyclegan generator test with device masks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3res; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_generators_gpu.py tests/test_bf16_gpu.py -m gpu -q -p no:cacheprovider -k "cyclegan or conv3x3_c64 or generator" > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -4 $O/tests.log
python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-180 $O/bench_bf16.json
python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; cut -c1-180 $O/bench_c5.json
bash scripts/gpu_prof_bench.sh r3res --dtype bf16 > $O/prof.log 2>&1; grep -E "conv3x3_c64_bf16_kernel|total" $O/prof.log
bash scripts/gpu_prof_bench.sh r3res_c5 --config c5 > $O/prof_c5.log 2>&1; grep -E "conv3x3_c64|convt|c256to3|total" $O/prof_c5.log
