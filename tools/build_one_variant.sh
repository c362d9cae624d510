#!/bin/bash
# Link a variant of the library that differs from the product build in ONE translation unit:
#   bash tools/build_one_variant.sh UNIT NAME path/to/variant/csrc     -> duodiff_amd/libduodiff_NAME.so   (the other objects from build/obj)
set -e
cd "$(dirname "$0")/.."
U=$1; N=$2; SRC=$3
mkdir -p build/obj_$N
/opt/rocm/bin/hipcc -O3 -std=c++20 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -Iinclude -c $SRC/$U.hip -o build/obj_$N/$U.o
objs=$(ls build/obj/*.o | grep -v "/$U.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc $objs build/obj_$N/$U.o -o duodiff_amd/libduodiff_$N.so
echo built duodiff_amd/libduodiff_$N.so
