#!/bin/bash
# A/B of the bf16 trunk convolution: v2 (default) against v1 (VCG_CONV3X3_V1=1): parity tests, then the kernel bench of both
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "conv3x3_c64_bf16 and not wgrad" > gpurun_out/v2_tests.log 2>&1
rc=$?; echo "v2 tests exit=$rc"; tail -5 gpurun_out/v2_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python scripts/kbench_bf16.py 32 > gpurun_out/v2_kbench.txt 2>&1 || exit 1
head -4 gpurun_out/v2_kbench.txt
VCG_CONV3X3_V1=1 timeout -k 10 200 python scripts/kbench_bf16.py 32 > gpurun_out/v1_kbench.txt 2>&1 || exit 1
head -4 gpurun_out/v1_kbench.txt
