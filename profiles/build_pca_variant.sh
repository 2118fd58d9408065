#!/bin/bash
# A variant of libpyvb_hip.so that differs in k_pca.hip only (compile-time switches of the VB-PCA sweep), into build/variants/:
#   bash profiles/build_pca_variant.sh stamp "-DP12_STAMP"     then on the GPU box   PYVB_HIP_LIB=build/variants/libpyvb_hip_stamp.so python ...
# The other objects are the shipped library's (make -C pyvb_amd/csrc first).
set -e
cd "$(dirname "$0")/../pyvb_amd/csrc"
make -j8 > /dev/null
mkdir -p ../../build/variants
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c k_pca.hip -o ../../build/variants/k_pca_$tag.o
  objs=$(ls *.o | grep -v '^k_pca.o$')
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs ../../build/variants/k_pca_$tag.o -o ../../build/variants/libpyvb_hip_$tag.so -ldl
  echo built $tag
done
