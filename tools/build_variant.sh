#!/bin/bash
# build_variants/<name>.so with extra -D flags: tools/build_variant.sh name -DMPV_WAVES=4 ...
set -e
name=$1; shift
P="unsupervised-probing-using-generative-diffusion-models_amd"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -shared -I include -I $P/csrc "$@" \
  -o build_variants/$name.so $P/csrc/mpv_api.hip $P/csrc/nsdiff_chain.hip $P/csrc/nsdiff_chain_f16.hip $P/csrc/sigma_head.hip $P/csrc/mpv_reduce.hip $P/csrc/biomass_sim.hip
echo built build_variants/$name.so
