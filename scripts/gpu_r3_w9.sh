# round 3: final/conv's bf16 weight gradient with the operand reads of the next k-step behind the MFMAs of this one
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -x -q -k "final_conv9x9_bf16_wgrad or train_step or trunk_generator or upsampling" > gpurun_out/w9_tests.log 2>&1; rc=$?; tail -3 gpurun_out/w9_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/w9_c3.json 2> gpurun_out/w9_c3.err && tail -1 gpurun_out/w9_c3.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 > gpurun_out/w9_c4.json 2> gpurun_out/w9_c4.err && tail -1 gpurun_out/w9_c4.json | cut -c1-200 &&
bash scripts/gpu_prof_bench.sh w9_bf16 --dtype bf16 | grep -E "gwgrad|wgrad9|wgrad3x3|total ms|rocprof"
