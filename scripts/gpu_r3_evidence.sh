# round-3 evidence pass: touched test files, the bf16 lines, step-level PMC traffic for C3 shard / C4 frame / C5 (merged into profiles/pmc_traffic.json
# by the caller from gpurun_out/), kernel-stats profiles
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3ev; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py tests/test_golden.py tests/test_model_gpu.py -m gpu -q -p no:cacheprovider --durations=8 > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -16 $O/tests.log
python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-180 $O/bench_bf16.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-180 $O/bench_c4.json
python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; cut -c1-180 $O/bench_c5.json
python bench.py --dtype bf16 --disc simple --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_simple.json 2> $O/bench_bf16_simple.err; echo "bench bf16 simple exit=$?"; cut -c1-180 $O/bench_bf16_simple.json
bash scripts/gpu_prof_bench.sh r3ev_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; tail -3 $O/prof_bf16.log
bash scripts/pmc_step.sh c4 2 --dtype bf16 --lr-size 540 --lr-width 960 --batch 4
bash scripts/pmc_step.sh c3 2 --dtype bf16
bash scripts/pmc_step.sh c5 2 --config c5
find gpurun_out/pmcstep_c4 gpurun_out/pmcstep_c3 gpurun_out/pmcstep_c5 -name "*kernel_trace.csv" -delete
du -sh gpurun_out/pmcstep_* | tail -3
