#!/bin/bash
# A variant of libpyvb_hip.so that differs in ONE translation unit (compile-time switches), into build/variants/:
#   bash profiles/build_variant.sh k_big gy3 "-DGY_SETS3"     then on the GPU box   PYVB_HIP_LIB=build/variants/libpyvb_hip_gy3.so python ...
# The other objects are the shipped library's (make -C pyvb_amd/csrc first).
set -e
cd "$(dirname "$0")/../pyvb_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../build/variants
unit=$1; tag=$2; flags=$3
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $unit.hip -o ../../build/variants/${unit}_$tag.o
objs=$(ls *.o | grep -v "^$unit.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../build/variants/${unit}_$tag.o -o ../../build/variants/libpyvb_hip_$tag.so -ldl
echo built $tag
