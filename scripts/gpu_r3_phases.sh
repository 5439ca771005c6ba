# round 3: the four output phases of a stride-2 data gradient as one launch of the LDS-tiled generic convolution: parity, A/B rates, C3 / C4 lines
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_bf16_gpu.py tests/test_fullsize_bf16_gpu.py -m gpu -x -q -k "generic_conv or discriminator or train_step or trunk_generator or upsampling or fullsize_generic or critics or transpose or first_conv" > gpurun_out/ph_tests.log 2>&1; rc=$?; tail -3 gpurun_out/ph_tests.log; [ $rc -eq 0 ] || exit $rc
echo "--- merged" > gpurun_out/ph_kbench.txt; timeout -k 10 300 python scripts/kbench_gconv.py >> gpurun_out/ph_kbench.txt 2>&1 &&
echo "--- one launch per phase (VCG_GCONV_MERGE_PHASES=0)" >> gpurun_out/ph_kbench.txt && VCG_GCONV_MERGE_PHASES=0 timeout -k 10 300 python scripts/kbench_gconv.py >> gpurun_out/ph_kbench.txt 2>&1 &&
echo "--- merged, C4 sizes" >> gpurun_out/ph_kbench.txt && timeout -k 10 300 python scripts/kbench_gconv.py c4 >> gpurun_out/ph_kbench.txt 2>&1 &&
echo "--- one launch per phase, C4 sizes" >> gpurun_out/ph_kbench.txt && VCG_GCONV_MERGE_PHASES=0 timeout -k 10 300 python scripts/kbench_gconv.py c4 >> gpurun_out/ph_kbench.txt 2>&1; grep -v amdgpu gpurun_out/ph_kbench.txt | cut -c1-110
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/ph_c3.json 2> gpurun_out/ph_c3.err && tail -1 gpurun_out/ph_c3.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --dtype bf16 --no-cpu-baseline --lr-size 540 --lr-width 960 --batch 4 --steps 10 --warmup 3 > gpurun_out/ph_c4.json 2> gpurun_out/ph_c4.err && tail -1 gpurun_out/ph_c4.json | cut -c1-200
