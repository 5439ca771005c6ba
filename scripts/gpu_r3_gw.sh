# round 3: generic bf16 weight gradient after the staging rewrite (slot arithmetic hoisted, buffer descriptors, ring of three stages):
# parity of everything that runs it, per-kernel rates, the C3 / C4 lines and the kernel breakdown of the C3 step
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -x -q -k "generic_conv or discriminator or train_step or trunk_generator or upsampling or fullsize_generic or critics" > gpurun_out/gw_tests.log 2>&1; rc=$?; tail -3 gpurun_out/gw_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python scripts/kbench_gconv.py > gpurun_out/gw_kbench_c3.txt 2>&1 && cat gpurun_out/gw_kbench_c3.txt &&
timeout -k 10 300 python scripts/kbench_gconv.py c4 > gpurun_out/gw_kbench_c4.txt 2>&1 && cat gpurun_out/gw_kbench_c4.txt &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/gw_c3.json 2> gpurun_out/gw_c3.err && tail -1 gpurun_out/gw_c3.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 > gpurun_out/gw_c4.json 2> gpurun_out/gw_c4.err && tail -1 gpurun_out/gw_c4.json | cut -c1-200 &&
bash scripts/gpu_prof_bench.sh gw_bf16 --dtype bf16 | grep -E "gwgrad|wgrad9|total ms|rocprof"
