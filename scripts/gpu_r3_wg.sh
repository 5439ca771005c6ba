# round 3: bf16 trunk weight gradient with the next tile's DMA pieces issued in front of k-steps 0..4: stamps, parity, C3 / C4 lines, kernel stats
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python scripts/micro/wg_stamps.py 8 256 256 > gpurun_out/wg_stamps_spread.txt 2>&1 && python scripts/micro/wg_stamps.py 4 540 960 >> gpurun_out/wg_stamps_spread.txt 2>&1 && cat gpurun_out/wg_stamps_spread.txt &&
timeout -k 10 800 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -x -q -k "wgrad or train_step or trunk" > gpurun_out/wg_tests.log 2>&1; rc=$?; tail -3 gpurun_out/wg_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/wg_c3.json 2> gpurun_out/wg_c3.err && tail -1 gpurun_out/wg_c3.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 > gpurun_out/wg_c4.json 2> gpurun_out/wg_c4.err && tail -1 gpurun_out/wg_c4.json | cut -c1-200 &&
bash scripts/gpu_prof_bench.sh wg_bf16 --dtype bf16 | grep -E "wgrad|total ms|rocprof"
