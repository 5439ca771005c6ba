# PatchGAN head on bf16 NHWC: kernel parity, the bf16 discriminator / train-step tests, A/B bench
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3head; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "cout1 or discriminator or train_step" > $O/t1.log 2>&1; echo "tests exit=$?"; tail -5 $O/t1.log
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err; echo "bench bf16 exit=$?"; cut -c1-200 $O/bench_bf16.json
VCG_HEAD_BF16=0 python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_off.json 2> $O/bench_bf16_off.err; echo "bench bf16 (fp32 head) exit=$?"; cut -c1-200 $O/bench_bf16_off.json
bash scripts/gpu_prof_bench.sh r3head_bf16 --dtype bf16 > $O/prof_bf16.log 2>&1; grep -E "cout1|total" $O/prof_bf16.log
