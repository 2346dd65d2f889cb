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

// ---- transposed LDS reads (gfx950 ds_read_b64_tr_b16): operands whose contraction index is the SLOW dimension in memory.
// A 16-lane group reads a 4-row x 16-column block of 16-bit elements and every lane receives one column of it: lane 4q+p of the
// group supplies the address of row q, columns 4p..4p+3 (8 bytes, 8-byte aligned); lane i receives column i, rows 0..3.
// EXEC must be all ones.  Two reads (rows 4g.. and 16+4g.. of a 32-row step for group g) make one 16x16x32 operand; both
// operands of an MFMA must use the same row assignment.
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;

__device__ __forceinline__ v4i16 tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(p)); }

__device__ __forceinline__ bf16x8 tr_frag(v4i16 lo, v4i16 hi) {
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    v8i16 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

}  // namespace vlg
