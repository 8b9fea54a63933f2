#!/bin/bash
# build_variants/<name>.so with extra -D flags applied to the chain kernels (the other translation units are compiled once, without
# the flags, into build_variants/obj/): tools/build_variant.sh name -DMPV_WAVES=4 ...
set -e
name=$1; shift
P="unsupervised-probing-using-generative-diffusion-models_amd"
CC="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize -fPIC -I include -I $P/csrc"
mkdir -p build_variants/obj
for f in $P/csrc/*.hip; do
  b=$(basename $f .hip)
  case $b in nsdiff_chain|nsdiff_chain_f16|mpv_api) continue;; esac
  if [ ! -f build_variants/obj/$b.o ] || [ $f -nt build_variants/obj/$b.o ]; then $CC -c $f -o build_variants/obj/$b.o & fi
done
wait
for b in nsdiff_chain nsdiff_chain_f16 mpv_api; do $CC "$@" -c $P/csrc/$b.hip -o build_variants/obj/${b}_$name.o & done
wait
$CC -shared -o build_variants/$name.so build_variants/obj/*_$name.o $(ls build_variants/obj/*.o | grep -v -E "/(nsdiff_chain|nsdiff_chain_f16|mpv_api)_")
rm -f build_variants/obj/*_$name.o
echo built build_variants/$name.so
