#!/bin/bash
# Diagnostic build of the bf16 transposed convolution (convt3x3_c64_bf16_kernel) with in-kernel s_memtime stamps (bf16_conv.hip, VCG_CT_STAMPS);
# extra -D flags select variants (CT_NO_DEFER: stores issued behind their phase's MFMAs, the form before round 3).  scripts/micro/ct_stamps.py prints where a tile's cycles go.  Build here, run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
P=video-cycle_gan-upscaling_amd
mkdir -p $P/build
for v in "" "$@"; do
  name=libvcg_ct_stamps${v:+_$v}.so
  /opt/rocm/bin/hipcc -shared -fPIC -O3 --offload-arch=gfx950 -std=c++17 -DVCG_CT_STAMPS ${v:+-D$v} -I include -I $P/csrc -Wno-unused-value -Wno-c++20-extensions \
      $P/csrc/bf16_conv.hip -o $P/build/$name
  echo $P/build/$name
done
