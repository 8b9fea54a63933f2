#!/bin/bash
# build_variants/<name>.so with extra -D flags applied to dts.hip only (the other objects are reused from build_variants/obj/ or the in-tree build/)
set -e
name=$1; shift
P="unsupervised-probing-using-generative-diffusion-models_amd"
CC="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -I include -I $P/csrc"
mkdir -p build_variants/obj
$CC -mllvm -amdgpu-mfma-vgpr-form "$@" -c $P/csrc/dts.hip -o build_variants/obj/dts_$name.o
$CC -shared -o build_variants/$name.so build_variants/obj/dts_$name.o $(ls $P/build/*.o | grep -v "/dts.o")
rm -f build_variants/obj/dts_$name.o
echo built build_variants/$name.so
