#!/bin/bash
# tools/build_variant_gemm.sh NAME [-Dflags...] -- the built library with vlg_gemm.hip recompiled under extra flags (A/B timing of the split-K weight
# gradients): tools/variants/lib_NAME.so, selected with VLGAE_AMD_LIB.  Needs a built vlgae_amd/_lib (python -m vlgae_amd.build).
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc "$@" -c -x hip vlgae_amd/csrc/vlg_gemm.hip -o tools/variants/gemm_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC $(ls vlgae_amd/_lib/*.o | grep -v vlg_gemm.o) tools/variants/gemm_$name.o -o tools/variants/lib_$name.so
echo tools/variants/lib_$name.so
