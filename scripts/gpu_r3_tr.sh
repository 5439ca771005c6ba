# line-contiguous epilogue of the 3-channel convolution kernel: parity, stamps, A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3tr; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -q -x -p no:cacheprovider -k "final_conv or first_conv or generator or train_step or edge_layers or upsampling or golden" > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -4 $O/tests.log
bash scripts/micro/i9_stamps_run.sh > $O/i9_stamps.txt 2>&1; grep -v amdgpu $O/i9_stamps.txt | grep -E "==|per launch|epilogue|MFMA loop|whole kernel"
for tr in 1 0; do
  VCG_I9_TR=$tr python bench.py --dtype bf16 --steps 30 --warmup 10 --no-cpu-baseline > $O/bench_bf16_tr$tr.json 2> $O/bench_bf16_tr$tr.err; echo "bench bf16 tr=$tr exit=$?"; cut -c1-180 $O/bench_bf16_tr$tr.json
  VCG_I9_TR=$tr python bench.py --dtype bf16 --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_c4_tr$tr.json 2> $O/bench_c4_tr$tr.err; echo "bench c4 tr=$tr exit=$?"; cut -c1-180 $O/bench_c4_tr$tr.json
done
VCG_I9_TR=1 bash scripts/gpu_prof_bench.sh r3tr1 --dtype bf16 > $O/prof1.log 2>&1; grep -E "c3to64_bf16_kernel|total" $O/prof1.log
