# iteration pass: GPU tests (bf16 + fp32 kernels + golden), headline C2 / C3-shard / C4 / C5 benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3iter; mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_bf16_gpu.py tests/test_kernels_gpu.py tests/test_golden.py tests/test_fullsize_identities_gpu.py -x -q -m gpu > $O/t1.log 2>&1; echo "tests exit=$?"; tail -3 $O/t1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 exit=$?"; cut -c1-200 $O/bench_c2.json
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-200 $O/bench_bf16.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-200 $O/bench_c4.json
python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; cut -c1-200 $O/bench_c5.json
bash scripts/gpu_prof_bench.sh r3iter_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; tail -3 $O/prof_bf16.log
bash scripts/gpu_prof_bench.sh r3iter_c2 > $O/prof_c2.log 2>&1; tail -30 $O/prof_c2.log
