#!/bin/bash
# tools/build_variant.sh NAME [-Dflags...] -- headline-only DP library variant for A/B timing (tools/variants/lib_NAME.so),
# selected at run time with VLGAE_AMD_LIB=tools/variants/lib_NAME.so.  Compiles in ~10 s instead of minutes.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffinite-math-only -DVLG_DP_HEADLINE_ONLY "$@" -shared \
  -x hip vlgae_amd/csrc/vlg_dp.hip vlgae_amd/csrc/vlg_capi.cpp -o tools/variants/lib_$name.so
echo tools/variants/lib_$name.so
