# round 3: fp32 path, normalisation statistics from the convolutions' epilogues: model-level parity, the C2 line with and without
set -o pipefail
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_generators_gpu.py tests/test_model_gpu.py tests/test_golden.py tests/test_fullsize_identities_gpu.py -m gpu -x -q > gpurun_out/f32stats_tests.log 2>&1; rc=$?; tail -3 gpurun_out/f32stats_tests.log; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/f32stats_c2.json 2> gpurun_out/f32stats_c2.err && tail -1 gpurun_out/f32stats_c2.json | cut -c1-200 &&
VCG_STATS_EPILOGUE_F32=0 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/f32stats_c2_off.json 2> gpurun_out/f32stats_c2_off.err && tail -1 gpurun_out/f32stats_c2_off.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/f32stats_c2b.json 2> gpurun_out/f32stats_c2b.err && tail -1 gpurun_out/f32stats_c2b.json | cut -c1-200 &&
timeout -k 10 300 python bench.py --no-cpu-baseline --disc simple --steps 10 --warmup 3 > gpurun_out/f32stats_simple.json 2> gpurun_out/f32stats_simple.err && tail -1 gpurun_out/f32stats_simple.json | cut -c1-200
