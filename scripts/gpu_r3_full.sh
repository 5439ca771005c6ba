# the whole GPU suite as the driver runs it (+ durations), then the chunked-tail A/B on C5 / C3 shard and fresh profiles
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3full; mkdir -p $O
rm -f gpurun_out/parity_report.txt
timeout -k 10 1000 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=30 > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -40 $O/tests.log
python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench c5 exit=$?"; cut -c1-180 $O/bench_c5.json
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-180 $O/bench_bf16.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-180 $O/bench_c4.json
bash scripts/micro/i9_stamps_run.sh > $O/i9_stamps.txt 2>&1; tail -30 $O/i9_stamps.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench c2 exit=$?"; cut -c1-180 $O/bench_c2.json
bash scripts/gpu_prof_bench.sh r3full_c5 --config c5 > $O/prof_c5.log 2>&1; tail -24 $O/prof_c5.log
bash scripts/gpu_prof_bench.sh r3full_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; tail -3 $O/prof_bf16.log
