# fused statistics / folded predict / bias-gradient records: bf16 tests, A/B bench, profile
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3stats; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu > $O/t1.log 2>&1; echo "tests exit=$?"; tail -5 $O/t1.log
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-200 $O/bench_bf16.json
VCG_FOLD_PREDICT=0 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err; echo "bench bf16 (predict unfolded) exit=$?"; cut -c1-200 $O/bench_bf16_off.json
python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench c4 exit=$?"; cut -c1-200 $O/bench_c4.json
bash scripts/gpu_prof_bench.sh r3stats_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; tail -34 $O/prof_bf16.log
