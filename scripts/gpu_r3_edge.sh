# bf16 edge layers (PatchGAN head, generator initial conv + PReLU): parity tests, A/B benches, profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3edge; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_dp_gpu.py -x -q -m gpu > $O/t1.log 2>&1; echo "tests exit=$?"; tail -5 $O/t1.log
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-200 $O/bench_bf16.json
VCG_FIRST_BF16=0 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err; echo "bench bf16 (fp32 critic block 1) exit=$?"; cut -c1-200 $O/bench_bf16_off.json
python bench.py --dtype bf16 --disc simple --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_simple.json 2> $O/bench_bf16_simple.err; echo "bench bf16 simple_512 exit=$?"; cut -c1-200 $O/bench_bf16_simple.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-200 $O/bench_c4.json
bash scripts/gpu_prof_bench.sh r3edge_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; grep -E "cout1|prelu_bwd|c3to64|nchw|act_bwd|total" $O/prof_bf16.log
