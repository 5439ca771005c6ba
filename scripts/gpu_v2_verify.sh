# bf16 trunk convolution v2: full bf16 parity suite, PMC counters of the bf16 kernels (batch 8), then the bf16 bench lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/v2
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/v2/tests_bf16.log 2>&1; rc=$?; echo "bf16 tests exit=$rc"; tail -3 gpurun_out/v2/tests_bf16.log
[ $rc -eq 0 ] || exit $rc
bash scripts/pmc_kbench.sh "8" kbench_bf16.py > gpurun_out/v2/pmc_bf16.log 2>&1 && python scripts/pmc_summary.py gpurun_out/pmc_8 > gpurun_out/v2/pmc_bf16_n8.txt 2>&1; grep "conv3x3_c64\|wgrad3x3_c64_bf16" gpurun_out/v2/pmc_bf16_n8.txt
for args in "--config c5" "--dtype bf16"; do
  tag=$(echo "$args" | tr -c "A-Za-z0-9\n" "_")
  python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/v2/bench$tag.json 2> gpurun_out/v2/bench$tag.err; echo "bench [$args] exit=$?"; cut -c1-200 gpurun_out/v2/bench$tag.json
done
