# fp32 trunk weight gradient on 4-row tiles: parity, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3wg; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_identities_gpu.py tests/test_golden.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -4 $O/tests.log
for t in 1 0; do
  VCG_WGRAD_TH4=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_th$t.json 2> $O/bench_c2_th$t.err; echo "bench c2 th4=$t exit=$?"; cut -c1-180 $O/bench_c2_th$t.json
done
VCG_WGRAD_TH4=1 bash scripts/gpu_prof_bench.sh r3wg1 > $O/prof1.log 2>&1; grep -E "wgrad_kernel<1, 4, 9|total" $O/prof1.log
VCG_WGRAD_TH4=0 bash scripts/gpu_prof_bench.sh r3wg0 > $O/prof0.log 2>&1; grep -E "wgrad_kernel<1, 4, 9|total" $O/prof0.log
