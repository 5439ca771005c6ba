# the N > 1 code path of bench.py over a real 1-rank RCCL communicator (fp32 headline and bf16 with the reference's default critic / losses)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3rccl1; mkdir -p $O
VCG_BENCH_FORCE_GROUP=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_rccl1.json 2> $O/bench_c2_rccl1.err; echo "c2 rccl1 exit=$?"; cut -c1-150 $O/bench_c2_rccl1.json; tail -2 $O/bench_c2_rccl1.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 exit=$?"; cut -c1-150 $O/bench_c2.json
VCG_BENCH_FORCE_GROUP=1 python bench.py --dtype bf16 --disc simple --gan-losses rel --disc-activation bi-log --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_rel_rccl1.json 2> $O/bench_bf16_rel_rccl1.err; echo "bf16 rel rccl1 exit=$?"; cut -c1-150 $O/bench_bf16_rel_rccl1.json; tail -2 $O/bench_bf16_rel_rccl1.err
python bench.py --dtype bf16 --disc simple --gan-losses rel --disc-activation bi-log --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_bf16_rel.json 2> $O/bench_bf16_rel.err; echo "bf16 rel exit=$?"; cut -c1-150 $O/bench_bf16_rel.json
