#!/bin/bash
# timing experiments: builds libkpl variants with parts of the score kernel compiled out
# (KPL_ABLATE: 1 = no forest walk, 8 = histogram update reduced to one add, 16 = score := cycles the
# wave lived, 64 = score := iterations of the feature loop) into build/ablate/
set -e
cd "$(dirname "$0")/.."
mkdir -p build/ablate
C=keypoint-learning_amd/csrc
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -DKPL_ABLATE=$v -x hip -c $C/kernels.hip -o build/ablate/kernels_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/ablate/libkpl_ab$v.so build/ablate/kernels_$v.o $C/api.o $C/forest.o -lz -Wl,-rpath,/opt/rocm/lib
done
