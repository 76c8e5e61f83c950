#!/bin/bash
# Development: builds experimental variants of the library, exp/libsxe_<k>.so with -DSXE_EXP=<k>
# (k = 0 is the tree as it is).  usage: bash tools/build_exp.sh 0 1 2 ...
cd "$(dirname "$0")/.." || exit 1
mkdir -p exp
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -Wall -Wno-unused-function -DSXE_EXP=$k \
      -o exp/libsxe_$k.so renderer_amd/csrc/sxe_api.hip renderer_amd/csrc/sxe_geometry.hip renderer_amd/csrc/sxe_raster.hip &
done
wait
ls -la exp/
