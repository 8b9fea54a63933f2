#!/bin/bash
# headline chain kernel, variants in build_variants/: time (181 windows) and accuracy against the exact kernel; tr_* variants: phase trace
for v in "$@"; do
  case $v in
    tr_*) TRACE=1 MPV_LIB=build_variants/$v.so python tools/chain_microbench.py 16 2 2>&1 | grep -v amdgpu.ids ;;
    *) ACC=1 MPV_LIB=build_variants/$v.so python tools/chain_microbench.py 181 2 2>&1 | grep -v amdgpu.ids ;;
  esac
done
