#!/bin/bash
# tools/build_variant_ffgemm.sh NAME [-Dflags...] -- the built library with vlg_ffgemm.hip recompiled under extra flags (A/B timing of the fused
# feed-forward layers): tools/variants/lib_NAME.so, selected with VLGAE_AMD_LIB.  Needs a built vlgae_amd/_lib (python -m vlgae_amd.build).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c -x hip vlgae_amd/csrc/vlg_ffgemm.hip -o tools/variants/ffgemm_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC $(ls vlgae_amd/_lib/*.o | grep -v vlg_ffgemm.o) tools/variants/ffgemm_$name.o -o tools/variants/lib_$name.so
echo tools/variants/lib_$name.so
