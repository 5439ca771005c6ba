# round 3, final evidence pass: the whole GPU suite as the driver runs it, every bench line, kernel-stats profiles, step-level PMC traffic
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -22 $O/tests.log
python bench.py --steps 20 --warmup 5 > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 exit=$?"; tail -1 $O/bench_c2.json | cut -c1-180
python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; tail -1 $O/bench_bf16.json | cut -c1-180
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; tail -1 $O/bench_c4.json | cut -c1-180
python bench.py --config c5 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; tail -1 $O/bench_c5.json | cut -c1-180
python bench.py --dtype bf16 --disc simple --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_simple.json 2> $O/bench_bf16_simple.err; echo "bench bf16 simple exit=$?"; tail -1 $O/bench_bf16_simple.json | cut -c1-180
python bench.py --dtype bf16 --fused-step --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_fused.json 2> $O/bench_bf16_fused.err; echo "bench bf16 fused exit=$?"; tail -1 $O/bench_bf16_fused.json | cut -c1-180
bash scripts/gpu_prof_bench.sh r3final_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; tail -3 $O/prof_bf16.log
bash scripts/pmc_step.sh c4 2 --dtype bf16 --lr-size 540 --lr-width 960 --batch 4
bash scripts/pmc_step.sh c3 2 --dtype bf16
python scripts/pmc_step_summary.py gpurun_out/pmcstep_c3 n8_256x256 3 8 x > $O/pmcstep_c3.txt 2>&1; tail -2 $O/pmcstep_c3.txt
python scripts/pmc_step_summary.py gpurun_out/pmcstep_c4 n4_540x960 3 4 x --alg-gb-per-frame 32 > $O/pmcstep_c4.txt 2>&1; tail -2 $O/pmcstep_c4.txt
find gpurun_out/pmcstep_c4 gpurun_out/pmcstep_c3 -name "*kernel_trace.csv" -delete
find gpurun_out/prof_r3final_bf16 -name "*kernel_trace.csv" -delete
du -sh gpurun_out/pmcstep_* | tail -3
