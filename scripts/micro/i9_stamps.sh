#!/bin/bash
# Diagnostic build of the 3-channel bf16 convolution (conv_c3to64_bf16_kernel: initial/conv forward, final/conv data gradient) with in-kernel
# s_memtime stamps (bf16_conv.hip, VCG_I9_STAMPS); scripts/micro/i9_stamps.py prints where a tile's cycles go.  Build here, run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
P=video-cycle_gan-upscaling_amd
mkdir -p $P/build
/opt/rocm/bin/hipcc -shared -fPIC -O3 --offload-arch=gfx950 -std=c++17 -DVCG_I9_STAMPS -I include -I $P/csrc -Wno-unused-value -Wno-c++20-extensions \
    $P/csrc/bf16_conv.hip -o $P/build/libvcg_i9_stamps.so
echo $P/build/libvcg_i9_stamps.so
