cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "bf16_discriminator or all_bf16 or trunk_generator_training" > gpurun_out/r2_d_tests.log 2>&1; tail -30 gpurun_out/r2_d_tests.log; grep "bf16 discriminator\|all-bf16\|C4 frame size, all" gpurun_out/parity_report.txt | tail -16
