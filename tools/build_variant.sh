#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags...]: builds build/variants/libkpl_<name>.so from
# build/variants/<name>/kernels.hip (a patched scratch copy; everything else from csrc/) -- timing experiments only,
# never shipped (build/ is git-ignored; the .so files travel to the GPU box).  Use with
#   KPL_LIB_PATH=build/variants/libkpl_<name>.so python bench.py --lean --no-parity ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; shift
V=$R/build/variants/$name
C=$R/keypoint-learning_amd/csrc
mkdir -p $V
[ -f $V/kernels.hip ] || cp $C/kernels.hip $V/kernels.hip
cp $C/kernels.h $C/forest.h $C/organized_normals.h $C/exact_math.h $C/soft_pair.h $V/ 2>/dev/null || true
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-result -x hip -I$C -I$R/include"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $V/kernels.hip -o $V/kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build/variants/libkpl_$name.so $V/kernels.o $C/organized_normals.o $C/api.o $C/forest.o -lz -Wl,-rpath,/opt/rocm/lib
echo built build/variants/libkpl_$name.so
