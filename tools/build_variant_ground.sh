#!/bin/bash
# tools/build_variant_ground.sh NAME [-Dflags...] -- grounding-loss-only library variant (vlg_ground + vlg_align) for A/B timing
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -shared \
  -x hip vlgae_amd/csrc/vlg_ground.hip vlgae_amd/csrc/vlg_align.hip vlgae_amd/csrc/vlg_dp.hip vlgae_amd/csrc/vlg_capi.cpp -DVLG_DP_HEADLINE_ONLY -o tools/variants/lib_$name.so
echo tools/variants/lib_$name.so
