// vlg_mfma.h -- matrix-core operand conventions shared by the alignment and arc-encoder kernels (gfx950).
//
// Both kernels contract over a dimension that is contiguous in memory on BOTH operands, which is the MFMA fragment order:
// lane l of a 16-row operand tile holds elements [row l & 15][k0 + EPL * (l >> 4) .. + EPL) -- one 16-byte read.
//   bf16 in: v_mfma_f32_16x16x32_bf16, K chunk 32, 8 elements per lane; fp32 in: four v_mfma_f32_16x16x4_f32 per chunk
//   of 16 (exact fp32 products; the K order inside a chunk is permuted identically on both operands).
// Accumulator / result tile: lane l, register n  <->  row 4 * (l >> 4) + n, column l & 15.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool F32IN>
struct MfmaCfg;
template <>
struct MfmaCfg<false> {      // bf16 in: K-chunks of 32, 6 row tiles (96 queries) of A fragments resident
    using T = uint16_t;
    using Frag = bf16x8;
    static constexpr int KW = 32, RTB = 6, EPL = 8;   // EPL = elements per lane per chunk (16 bytes)
};
template <>
struct MfmaCfg<true> {       // fp32 in: K-chunks of 16, 3 row tiles (48 queries)
    using T = float;
    using Frag = f32x4;
    static constexpr int KW = 16, RTB = 3, EPL = 4;
};

template <bool F32IN>
__device__ __forceinline__ f32x4 mma_chunk(const typename MfmaCfg<F32IN>::Frag& a,
                                           const typename MfmaCfg<F32IN>::Frag& b, f32x4 acc) {
    if constexpr (F32IN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc, 0, 0, 0);
        return acc;
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
    }
}

}  // namespace vlg
