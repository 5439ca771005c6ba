#!/bin/bash
# Diagnostic build of the bf16 trunk weight gradient (wgrad3x3_c64_bf16_kernel) with in-kernel s_memtime stamps (bf16_wgrad.hip, VCG_WG_STAMPS);
# extra -D flags select variants.  scripts/micro/wg_stamps.py prints where a tile's cycles go.  Build here, run on the GPU box.
set -e
cd "$(dirname "$0")/../.."
P=video-cycle_gan-upscaling_amd
mkdir -p $P/build
for v in "" "$@"; do
  name=libvcg_wg_stamps${v:+_$v}.so
  /opt/rocm/bin/hipcc -shared -fPIC -O3 --offload-arch=gfx950 -std=c++17 -DVCG_WG_STAMPS ${v:+-D$v} -I include -I $P/csrc -Wno-unused-value -Wno-c++20-extensions \
      $P/csrc/bf16_wgrad.hip -o $P/build/$name
  echo $P/build/$name
done
