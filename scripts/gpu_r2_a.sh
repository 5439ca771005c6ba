set -o pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_model_gpu.py -m gpu -q -k "reference_default" > gpurun_out/r2_a_tests2.log 2>&1; tail -5 gpurun_out/r2_a_tests2.log
python scripts/kbench.py > gpurun_out/r2_a_kbench.txt 2>&1; cat gpurun_out/r2_a_kbench.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r2_a_bench.json 2> gpurun_out/r2_a_bench.err; cat gpurun_out/r2_a_bench.json
