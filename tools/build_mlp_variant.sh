#!/bin/bash
# Link a variant of the library that differs from the product build in mlp_fused.hip only:
#   bash tools/build_mlp_variant.sh NAME path/to/variant/csrc     -> duodiff_amd/libduodiff_NAME.so   (product objects from build/obj)
set -e
cd "$(dirname "$0")/.."
N=$1; SRC=$2
mkdir -p build/obj_$N
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -Iinclude -c $SRC/mlp_fused.hip -o build/obj_$N/mlp_fused.o
objs=$(ls build/obj/*.o | grep -v mlp_fused.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs build/obj_$N/mlp_fused.o -o duodiff_amd/libduodiff_$N.so
echo built duodiff_amd/libduodiff_$N.so
