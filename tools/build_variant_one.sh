#!/bin/bash
# build_variants/<name>.so with extra flags applied to ONE translation unit (the other objects come from the in-tree build/):
#   tools/build_variant_one.sh stg name -mllvm -amdgpu-mfma-vgpr-form
set -e
unit=$1; name=$2; shift 2
P="unsupervised-probing-using-generative-diffusion-models_amd"
CC="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -I include -I $P/csrc"
mkdir -p build_variants/obj
extra=""; [ $unit = dts ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
$CC $extra "$@" -c $P/csrc/$unit.hip -o build_variants/obj/${unit}_$name.o
$CC -shared -o build_variants/$name.so build_variants/obj/${unit}_$name.o $(ls $P/build/*.o | grep -v "/$unit.o")
rm -f build_variants/obj/${unit}_$name.o
echo built build_variants/$name.so
