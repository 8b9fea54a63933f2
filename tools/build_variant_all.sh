#!/bin/bash
# build_variants/<name>.so of the WHOLE library with extra -D flags (kernel-variant experiments; load with MPV_LIB=build_variants/<name>.so):
#   tools/build_variant_all.sh name -DTCN_PREFETCH=1 ...
set -e
name=$1; shift
P="unsupervised-probing-using-generative-diffusion-models_amd"
mkdir -p build_variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared -I include -I $P/csrc "$@" -o build_variants/$name.so $P/csrc/*.hip
echo built build_variants/$name.so
