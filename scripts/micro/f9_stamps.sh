#!/bin/bash
# Diagnostic build of the bf16 final convolution (9x9, 256 -> 3) with in-kernel s_memtime stamps (bf16_conv.hip, VCG_F9_STAMPS);
# scripts/micro/f9_stamps.py prints where an input row's cycles go.  Build here (hipcc cross-compiles), run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
P=video-cycle_gan-upscaling_amd
mkdir -p $P/build
/opt/rocm/bin/hipcc -shared -fPIC -O3 --offload-arch=gfx950 -std=c++17 -DVCG_F9_STAMPS -I include -I $P/csrc -Wno-unused-value \
    $P/csrc/bf16_conv.hip -o $P/build/libvcg_f9_stamps.so
echo $P/build/libvcg_f9_stamps.so
