cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
# rehearsal of the N-rank launcher path on one GPU (gloo ranks sharing cuda:0): numbers are meaningless, the code path is what is checked
VCG_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 2 --lr-size 64 --res-blocks 2 > gpurun_out/r2_dp_rehearsal.log 2>&1; echo "rehearsal wass exit=$?"; tail -2 gpurun_out/r2_dp_rehearsal.log | cut -c1-600
VCG_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 3 --warmup 1 --batch 2 --lr-size 64 --res-blocks 2 --gan-losses rel --disc simple --disc-activation bi-log --dtype bf16 > gpurun_out/r2_dp_rehearsal2.log 2>&1; echo "rehearsal rel/bf16 exit=$?"; tail -2 gpurun_out/r2_dp_rehearsal2.log | cut -c1-600
# the real launcher line with ONE rank over RCCL (what the driver runs with N ranks)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2_dp_n1.log 2>&1; echo "torchrun n=1 exit=$?"; tail -1 gpurun_out/r2_dp_n1.log | cut -c1-300
timeout -k 10 300 python scripts/dp_nccl_smoke.py > gpurun_out/r2_dp_nccl_smoke.log 2>&1; echo "nccl smoke exit=$?"; tail -4 gpurun_out/r2_dp_nccl_smoke.log
