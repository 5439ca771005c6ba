#!/bin/bash
# Diagnostic build of the bf16 trunk convolution with in-kernel s_memtime stamps (bf16_conv.hip, VCG_V2_STAMPS), then
# scripts/micro/v2_stamps.py prints where a tile's cycles go.  Build here (hipcc cross-compiles), run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
P=video-cycle_gan-upscaling_amd
mkdir -p $P/build
/opt/rocm/bin/hipcc -shared -fPIC -O3 --offload-arch=gfx950 -std=c++17 -DVCG_V2_STAMPS -I include -I $P/csrc -Wno-unused-value \
    $P/csrc/bf16_conv.hip -o $P/build/libvcg_v2_stamps.so
echo $P/build/libvcg_v2_stamps.so
