# round-3 baseline at the round-2 kernels: fresh bf16 / fp32 step profiles, step-level PMC traffic for C3 shard / C4 frame / C5
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3base; mkdir -p $O
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-200 $O/bench_bf16.json
bash scripts/gpu_prof_bench.sh r3base_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; tail -32 $O/prof_bf16.log
bash scripts/gpu_prof_bench.sh r3base_c2 > $O/prof_c2.log 2>&1; tail -3 $O/prof_c2.log
bash scripts/pmc_step.sh c4 2 --dtype bf16 --lr-size 540 --lr-width 960 --batch 4
python scripts/pmc_step_summary.py gpurun_out/pmcstep_c4 n4_540x960 3 4 r3base --alg-gb-per-frame 32 > $O/pmcstep_c4.txt 2>&1; tail -4 $O/pmcstep_c4.txt
bash scripts/pmc_step.sh c3 2 --dtype bf16
python scripts/pmc_step_summary.py gpurun_out/pmcstep_c3 n8_256x256 3 8 r3base > $O/pmcstep_c3.txt 2>&1; tail -2 $O/pmcstep_c3.txt
bash scripts/pmc_step.sh c5 2 --config c5
python scripts/pmc_step_summary.py gpurun_out/pmcstep_c5 n32_256x256 6 32 r3base > $O/pmcstep_c5.txt 2>&1; tail -2 $O/pmcstep_c5.txt
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-200 $O/bench_c4.json
python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; cut -c1-200 $O/bench_c5.json
# remove the bulky raw traces, keep the counter CSVs small
find gpurun_out/pmcstep_c4 gpurun_out/pmcstep_c3 gpurun_out/pmcstep_c5 -name "*kernel_trace.csv" -delete
du -sh gpurun_out/pmcstep_* | tail -3
