#!/bin/bash
# Builds libpyvb_hip.so with other compile-time kernel variants into build/variants/ (A/B runs on the GPU box:
# PYVB_HIP_LIB=build/variants/libpyvb_hip_<tag>.so python bench.py ...).  usage: build_variants.sh tag "-DFLAGS" ...
set -e
cd "$(dirname "$0")/../pyvb_amd/csrc"
mkdir -p ../../build/variants
while [ $# -ge 2 ]; do
  tag=$1; flags=$2; shift 2
  d=../../build/variants/obj_$tag; mkdir -p $d
  for f in api k_prep k_sweep k_stats k_params k_cols k_wishart k_missing k_tape api_pca k_pca; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $flags -c $f.hip -o $d/$f.o &
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/*.o -o ../../build/variants/libpyvb_hip_$tag.so -ldl
  echo built $tag
done
