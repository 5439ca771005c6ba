# round 3: bf16 weight gradient of the 3-channel layers (initial/conv, the critics' block 1): model-level parity, C3 / C4 lines, kernel stats
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_model_gpu.py tests/test_dp_gpu.py -m gpu -x -q -k "bf16 or fused or graph" > gpurun_out/w3_tests.log 2>&1; rc=$?; tail -3 gpurun_out/w3_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/w3_c3.json 2> gpurun_out/w3_c3.err && tail -1 gpurun_out/w3_c3.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 > gpurun_out/w3_c4.json 2> gpurun_out/w3_c4.err && tail -1 gpurun_out/w3_c4.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --disc simple --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/w3_simple.json 2> gpurun_out/w3_simple.err && tail -1 gpurun_out/w3_simple.json | cut -c1-200 &&
bash scripts/gpu_prof_bench.sh w3_bf16 --dtype bf16 | grep -E "wgrad|prelu_bwd|pack_frames|nhwc_to_f32|total ms|rocprof"
