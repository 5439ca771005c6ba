# round-3 host-side changes: fused step, DP overlap + collective timing, bench rehearsal (2 gloo ranks on the one GPU)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3host; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -x -q -m gpu > $O/t1.log 2>&1; echo "tests exit=$?"; tail -5 $O/t1.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-step > $O/bench_c2_fused.json 2> $O/bench_c2_fused.err; echo "bench c2 fused exit=$?"; cut -c1-200 $O/bench_c2_fused.json
python bench.py --dtype bf16 --steps 20 --warmup 5 --no-cpu-baseline --fused-step > $O/bench_bf16_fused.json 2> $O/bench_bf16_fused.err; echo "bench bf16 fused exit=$?"; cut -c1-200 $O/bench_bf16_fused.json
VCG_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_rehearsal_dp2.json 2> $O/bench_rehearsal_dp2.err; echo "rehearsal exit=$?"; cat $O/bench_rehearsal_dp2.json
VCG_BENCH_REHEARSAL=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29545 bench.py --gpus 2 --steps 10 --warmup 3 --dtype bf16 --disc simple --gan-losses rel --disc-activation bi-log > $O/bench_rehearsal_dp2_bf16_rel.json 2> $O/bench_rehearsal_dp2_bf16_rel.err; echo "rehearsal bf16 rel exit=$?"; cut -c1-300 $O/bench_rehearsal_dp2_bf16_rel.json
