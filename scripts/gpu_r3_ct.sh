# transposed convolution with deferred stores: parity (kernel + model level + full-size adjoints), benches
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ct; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -q -x -p no:cacheprovider -k "transpose or generator or train_step or upsampling or golden or c5 or c4" > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -4 $O/tests.log
python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; cut -c1-180 $O/bench_c5.json
python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-180 $O/bench_bf16.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-180 $O/bench_c4.json
bash scripts/gpu_prof_bench.sh r3ct_c5 --config c5 > $O/prof_c5.log 2>&1; grep -E "convt|c256to3|conv3x3|total" $O/prof_c5.log
