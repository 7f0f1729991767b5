#!/bin/bash
# timing experiments: builds libkpl variants with parts of the score kernel compiled out
# (KPL_ABLATE bit 0 = no forest walk, bit 1 = no histogram accumulation) into build/ablate/
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ablate
C=keypoint-learning_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DKPL_ABLATE=$v -x hip -c $C/kernels.hip -o build/ablate/kernels_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/libkpl_ab$v.so build/ablate/kernels_$v.o $C/api.o $C/forest.o -lz -Wl,-rpath,/opt/rocm/lib
done
