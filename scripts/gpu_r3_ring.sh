# round 3: LDS-tiled generic convolution, 64-output-channel form with loader waves + a ring of three stages: the transposed convolution's backward, parity, rates, C3 / C4 / kernel stats
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 120 python scripts/micro/convt_bwd_bench.py > gpurun_out/ring_ctd.txt 2>&1 && cat gpurun_out/ring_ctd.txt &&
timeout -k 10 800 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -x -q -k "generic_conv or discriminator or train_step or trunk_generator or upsampling or fullsize_generic or critics or transpose or stats_epilogue" > gpurun_out/ring_tests.log 2>&1; rc=$?; tail -3 gpurun_out/ring_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/kbench_gconv.py > gpurun_out/ring_kbench_c3.txt 2>&1 && cat gpurun_out/ring_kbench_c3.txt &&
timeout -k 10 300 python scripts/kbench_gconv.py c4 > gpurun_out/ring_kbench_c4.txt 2>&1 && cat gpurun_out/ring_kbench_c4.txt &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ring_c3.json 2> gpurun_out/ring_c3.err && tail -1 gpurun_out/ring_c3.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 > gpurun_out/ring_c4.json 2> gpurun_out/ring_c4.err && tail -1 gpurun_out/ring_c4.json | cut -c1-200 &&
bash scripts/gpu_prof_bench.sh ring_bf16 --dtype bf16 | grep -E "gconv|total ms|rocprof"
