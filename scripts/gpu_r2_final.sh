# round-2 measurement pass: headline bench (+cpu baseline), rocprofv3 kernel stats, PMC traffic of the dominant kernels, secondary lines
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/bench_c2.json 2> gpurun_out/final/bench_c2.err; echo "bench c2 exit=$?"; cut -c1-260 gpurun_out/final/bench_c2.json
bash scripts/gpu_prof_bench.sh r2_final > gpurun_out/final/prof.log 2>&1; tail -3 gpurun_out/final/prof.log
bash scripts/pmc_kbench.sh "trunk 3x3" > gpurun_out/final/pmc_trunk.log 2>&1; python scripts/pmc_summary.py gpurun_out/pmc_trunk_3x3 > gpurun_out/final/pmc_trunk_conv.txt 2>&1; cat gpurun_out/final/pmc_trunk_conv.txt
bash scripts/pmc_kbench.sh "8" kbench_bf16.py > gpurun_out/final/pmc_bf16.log 2>&1; python scripts/pmc_summary.py gpurun_out/pmc_8 > gpurun_out/final/pmc_bf16_n8.txt 2>&1; grep "conv3x3_c64\|wgrad3x3_c64_bf16" gpurun_out/final/pmc_bf16_n8.txt
for args in "--disc simple" "--kernel-size 5" "--gan-losses rel --disc simple --disc-activation bi-log" "--dtype bf16" "--dtype bf16 --lr-size 540 --lr-width 960 --batch 4" "--lr-size 540 --lr-width 960 --batch 4" "--config c5"; do
  tag=$(echo "$args" | tr -c "A-Za-z0-9\n" "_")
  python bench.py $args --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/final/bench$tag.json 2> gpurun_out/final/bench$tag.err; echo "bench [$args] exit=$?"; cut -c1-200 gpurun_out/final/bench$tag.json
done
