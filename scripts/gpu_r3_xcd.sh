# XCD-grouped channel-block mapping (convT forward, 3-channel conv data gradient) A/B + the cyclegan generator test
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3xcd; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_generators_gpu.py tests/test_bf16_gpu.py -m gpu -q -x -p no:cacheprovider -k "cyclegan or transpose or final_conv or first_conv or train_step" > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -4 $O/tests.log
for g in 1 0; do
  VCG_XCD_GROUP=$g python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16_xcd$g.json 2> $O/bench_bf16_xcd$g.err; echo "bench bf16 xcd=$g exit=$?"; cut -c1-180 $O/bench_bf16_xcd$g.json
  VCG_XCD_GROUP=$g python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c5_xcd$g.json 2> $O/bench_c5_xcd$g.err; echo "bench c5 xcd=$g exit=$?"; cut -c1-180 $O/bench_c5_xcd$g.json
done
VCG_XCD_GROUP=1 bash scripts/gpu_prof_bench.sh r3xcd1 --dtype bf16 > $O/prof1.log 2>&1; grep -E "convt3x3|c3to64_bf16_kernel<9|total" $O/prof1.log
VCG_XCD_GROUP=0 bash scripts/gpu_prof_bench.sh r3xcd0 --dtype bf16 > $O/prof0.log 2>&1; grep -E "convt3x3|c3to64_bf16_kernel<9|total" $O/prof0.log
