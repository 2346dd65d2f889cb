// vlg_ffgemm.hip -- the parser feed-forwards' Linear layers over ALL token rows fused with the element-wise pass behind them
// (`DMVSkipConnectEncoder`, src/model/nn/dmv_spec.py:38-54: Linear -> [+ skip connection] -> LeakyReLU [-> nn.Dropout], five times per
// step on up to 4 (B L + T + 3) = 41 152 rows of H = 256 channels, and the same five backwards).
//
// As a library GEMM + vlg_ff_act pair every stage writes its pre-activation (21 MB in bf16), reads it back, and writes the
// activation: 4 passes over the activation where the mathematics needs 2, and the library's 256 x 192 tiles stream [41 152 x 256] x
// [256 x 256] at ~2.1 TB/s (profiles/r06_z_train_step_sequence.txt: 12-26 us per product + 6-12 us per element-wise pass).  Here the
// layer is a ROW-STREAMING product: K = 256, so the whole weight block [256 outputs x 256] of a workgroup lives in its wavefronts'
// REGISTERS as MFMA B fragments (wave w of eight: output columns 32 w .. 32 w + 31, 16 fragments of 8 bf16 = 64 VGPRs), the rows stream
// through -- 32-row tiles, global -> registers one tile ahead -> LDS -> A fragments (ds_read_b128, pitch 544 B: conflict-free) ->
// v_mfma_f32_16x16x32_bf16 -- and the fp32 accumulators go through an LDS tile into the SAME element-wise code the separate pass ran
// (vlg_ff.hip: eight channels of a row per thread, 16-byte loads / stores, one rounding per stored element): bias, skip connection,
// LeakyReLU, mask or counter-based dropout and the (dir,val) store permutation forward; LeakyReLU' of the stored activation, mask and
// the skip connection's group sum backward.  Bytes per stage: rows x 256 x 2 in + out (+ the residual / activation rows): the
// floor of the pair it replaces is twice that.
//
// bf16 storage only (fp32 activations keep the library path); H = 256 exactly (config/model/vlgae.yaml: the parser's hidden width).  The
// contraction is K = 256 forward; the backward launches also take K = 512 (the cotangent of a two-block stage: 128 weight VGPRs) and K = 32
// (the cotangent of the folded projections, 2 r = 32 columns, the weight in its own [k][n] layout).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "vlg_common.h"
#include "vlg_mfma.h"
#include "vlg_rng.h"
#include "vlg_rows.h"

namespace vlg {

namespace {

constexpr int kFgH = 256;                 // channels in and out per column block
constexpr int kFgRows = 32;               // rows per tile
constexpr int kFgThreads = 512;           // eight wavefronts: 32 output columns each (ONE workgroup per CU: its 128 KB weight block is read once per CU)
// bytes per LDS row of the input tile of K channels: K = 256 -> 544 (136 words = 8 mod 64), K = 512 -> 1056 (264 = 8 mod 64), K = 32 -> 96 (24 words):
// in each case the four 16-lane groups of a ds_read_b128 hit disjoint banks
constexpr int fg_pitch(int K) { return K * 2 + 32; }
constexpr int fg_lds(int K) { return 2 * kFgRows * fg_pitch(K); }   // two input images

struct FgArgs {
    const uint16_t* x;      // [rows][ldx] input rows (bf16), the first K channels
    int ldx;
    const uint16_t* w;      // [nb * 256][K]: w[n][k], the weight of output channel n (nn.Linear layout; a transposed copy for x @ W; K = 512: two
                            // [256][256] blocks) -- or, w_kn, [K][256]: w[k][n]
    int w_kn;
    int ldw, ncols;         // w_kn: elements between the rows of w, and its valid columns (columns past them read as zeros)
    int plain, ldo;         // plain: out[row][256 y + c] = bf16(acc + bias), rows ldo elements apart, columns < ncols only (no element-wise pass)
    const uint16_t* bias;   // [nb * 256] or null
    long long rows;
    float slope;
    // ---- forward epilogue: out[orow][c] = leaky(acc + bias[y 256 + c] + res[row >> rs][c]) * keep, orow = (row >> rs) om + y oy + (row & ((1 << rs) - 1))
    const uint16_t* res;    // [rows >> rs][256] or null
    int rs, om, oy;
    // ---- backward epilogue (bwd != 0): out[orow][c] = leaky'(act[row][c]) * acc * keep[row]; sum[m][c] (+)= sum_j stored(out), rows m J + j; orow = swap ? 4 m + swap2(j) : row
    int bwd, J, lj, swap, accumulate;
    const uint16_t* act;    // [rows][256] stored activations (backward)
    float* sum;             // [rows / J][256] fp32 or null
    // ---- keep: an explicit mask (times mask_scale) or the counter-based draw (rng), indexed by orow forward and by row backward (as vlg_ff_act*)
    const uint16_t* mask;
    float mask_scale;
    const uint64_t* rng;
    uint32_t site, thr;
    uint16_t* out;          // [.][256]
    // ---- K = 512 backward only: the head of the encoder (as vlg_ff_mlp_act_backward): out = leaky'(act) * keep * (add[row] + acc), keep = drop_head[row / L] for
    // rows < M0 (one mask per sentence) and drop_small[row - M0] behind them (one value per row); add null: none of it
    const float* add;       // [rows][256] fp32
    const float* drop_head; // [M0 / L][256] fp32 or null
    const float* drop_small;// [rows - M0] fp32 or null
    int M0, L;
};

// The element-wise pass's options as a compile-time word M (>= 0) or read from the argument block (M = -1, the generic image).  The options are
// uniform, but as RUN-TIME conditions they were ~140 scalar branches per 32-row tile (the unrolled (row tile, column tile) bodies each test them):
// a tile is ~6 500 cycles of which the products are 500, so the kernel family was bound by its own control flow.  fg_mode() names the combinations
// the training step launches; anything else takes the generic image.
enum : int { kMBwd = 1, kMRes = 2, kMMask = 4, kMRng = 8, kMSum = 16, kMSwap = 32, kMAcc = 64, kMPlain = 128, kMJ2 = 256, kMJ4 = 512, kMRs1 = 1024, kMAdd = 2048 };

__device__ __forceinline__ void unpack4(uint2 w, float (&v)[4]) {   // four bf16 in two words -> fp32
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
}

__device__ __forceinline__ int fg_swap2(int j) { return ((j & 1) << 1) | (j >> 1); }

// bf16 conversions on the hardware path (v_cvt_pk_bf16_f32, round to nearest even like vlg_rows.h's f2bf: the epilogue rounds 16 values per item)
__device__ __forceinline__ float fg_round(float v) { return (float)(__bf16)v; }
__device__ __forceinline__ void fg_store8(uint16_t* p, const float (&v)[8]) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 a = {(__bf16)v[0], (__bf16)v[1]}, b = {(__bf16)v[2], (__bf16)v[3]}, c = {(__bf16)v[4], (__bf16)v[5]}, d = {(__bf16)v[6], (__bf16)v[7]};
    uint4 w;
    w.x = __builtin_bit_cast(uint32_t, a); w.y = __builtin_bit_cast(uint32_t, b); w.z = __builtin_bit_cast(uint32_t, c); w.w = __builtin_bit_cast(uint32_t, d);
    *reinterpret_cast<uint4*>(p) = w;
}

__device__ __forceinline__ void fg_store4(uint16_t* p, const float (&v)[4]) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 a = {(__bf16)v[0], (__bf16)v[1]}, b = {(__bf16)v[2], (__bf16)v[3]};
    uint2 w;
    w.x = __builtin_bit_cast(uint32_t, a); w.y = __builtin_bit_cast(uint32_t, b);
    *reinterpret_cast<uint2*>(p) = w;
}
// sum over the J (1, 2 or 4) consecutive lanes of a row group (lanes r, r ^ 1, r ^ 2, r ^ 3 of a quad): every lane gets the group's sum
// (DPP quad permutations: one vector instruction each; __shfl_xor goes through the LDS crossbar)
__device__ __forceinline__ float fg_quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
}
__device__ __forceinline__ float fg_quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
}
__device__ __forceinline__ float fg_group_sum(float v, int J) {
    if (J >= 2) v += fg_quad_xor1(v);
    if (J >= 4) v += fg_quad_xor2(v);
    return v;
}
// The counter-based keep flags of a tile for this lane: km[rt][nt][k] for its four channels of the 8-channel groups (orow(rt), c >> 3).  The lane
// 16 further on holds the other half of the same groups: of the two, the one with even kg draws the groups of row tile 0 and the other those of
// row tile 1 (one Philox call per group and lane pair, as vlg_ff_act makes one per group), and they exchange the halves they owe each other.
__device__ __forceinline__ void fg_tile_keep(const FgArgs& a, long long krow0, long long krow1, int wave, int kg, int hv, int c0, float (&km)[2][2][4]) {
    const int mine = kg & 1;      // (hv: 8-channel groups per row of the tensor the draw is indexed over; c0: this workgroup's first column in it)
    const uint64_t seed = a.rng[0], step = a.rng[1];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int c = c0 + wave * 32 + nt * 16 + kg * 4;
        const uint64_t g = (uint64_t)(mine ? krow1 : krow0) * hv + (c >> 3);
        const uint4 q = philox4x32_10(make_uint4((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)step, (uint32_t)(step >> 32)),
                                      make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (a.site * 0x9E3779B9u)));
        // my half of my row tile's group; the partner's half (the other one) of it goes over
        const uint32_t own0 = mine ? q.z : q.x, own1 = mine ? q.w : q.y, give0 = mine ? q.x : q.z, give1 = mine ? q.y : q.w;
        const uint32_t got0 = (uint32_t)__shfl_xor((int)give0, 16, 64), got1 = (uint32_t)__shfl_xor((int)give1, 16, 64);
        // (named words, explicit selects: small arrays indexed by a lane-dependent flag were placed in scratch memory)
        const uint32_t a0 = mine ? got0 : own0, a1 = mine ? got1 : own1, b0 = mine ? own0 : got0, b1 = mine ? own1 : got1;   // row tile 0 | 1
        km[0][nt][0] = (a0 & 0xffffu) >= a.thr ? a.mask_scale : 0.f;
        km[0][nt][1] = (a0 >> 16) >= a.thr ? a.mask_scale : 0.f;
        km[0][nt][2] = (a1 & 0xffffu) >= a.thr ? a.mask_scale : 0.f;
        km[0][nt][3] = (a1 >> 16) >= a.thr ? a.mask_scale : 0.f;
        km[1][nt][0] = (b0 & 0xffffu) >= a.thr ? a.mask_scale : 0.f;
        km[1][nt][1] = (b0 >> 16) >= a.thr ? a.mask_scale : 0.f;
        km[1][nt][2] = (b1 & 0xffffu) >= a.thr ? a.mask_scale : 0.f;
        km[1][nt][3] = (b1 >> 16) >= a.thr ? a.mask_scale : 0.f;
    }
}

// One 32-row tile.  The product runs TRANSPOSED on the matrix cores -- D[channel][row] = W (A operand, from registers) x X^T (B operand, from the
// LDS image `xs`) -- so that a lane's accumulator registers are FOUR CONSECUTIVE CHANNELS of ONE ROW (lane l, register n <-> channel
// c0 + 4 (l >> 4) + n, row l & 15): the element-wise pass runs on the accumulators in registers, with 8-byte loads / stores of the row's
// other tensors, and nothing goes back through the LDS (the first version wrote the fp32 accumulators to an LDS tile and read them back
// row-major behind a barrier: eight waves in lock step, ~6 700 cycles per tile for 512 cycles of MFMA).
template <int KS, bool PRE = true, int M = -1>
__device__ __forceinline__ void fg_tile(const FgArgs& a, long long t, const char* xs, const bf16x8 (&wf)[KS][2], const float (&bias4)[2][4], int tid, int y,
                                        char* ys = nullptr) {      // ys: an LDS image (the input tile's layout) that takes the stored values as well -- the next stage's input
    constexpr int kFgXPitch = fg_pitch(KS * 32);
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, kg = lane >> 4;
    // the options: constants of this image (M >= 0) or the argument block's fields
    const bool f_bwd = M >= 0 ? (M & kMBwd) != 0 : a.bwd != 0, f_res = M >= 0 ? (M & kMRes) != 0 : a.res != nullptr;
    const bool f_mask = M >= 0 ? (M & kMMask) != 0 : a.mask != nullptr, f_rng = M >= 0 ? (M & kMRng) != 0 : a.rng != nullptr;
    const bool f_sum = M >= 0 ? (M & kMSum) != 0 : a.sum != nullptr, f_swap = M >= 0 ? (M & kMSwap) != 0 : a.swap != 0;
    const bool f_acc = M >= 0 ? (M & kMAcc) != 0 : a.accumulate != 0, f_plain = M >= 0 ? (M & kMPlain) != 0 : a.plain != 0;
    const bool f_add = M >= 0 ? (M & kMAdd) != 0 : a.add != nullptr;
    const int f_J = M >= 0 ? ((M & kMJ4) ? 4 : ((M & kMJ2) ? 2 : 1)) : a.J, f_lj = M >= 0 ? ((M & kMJ4) ? 2 : ((M & kMJ2) ? 1 : 0)) : a.lj;
    const int f_rs = M >= 0 ? ((M & kMRs1) ? 1 : 0) : a.rs;
    // The element-wise pass's row operand of this tile (backward: the stored activations; forward: the skip connection's rows) is requested HERE, ahead
    // of the products, and used behind them: issued inside the epilogue (under its row conditions) every tile paid its round trip in full, eight
    // waves in lock step.  Unconditional: a row past the end re-reads the last row, a stage without the operand re-reads its own input.
    const uint16_t* const ep = f_bwd ? a.act : (f_res ? a.res : a.x);
    const int ep_ld = (f_bwd || f_res) ? kFgH : a.ldx, ep_sh = f_bwd ? 0 : (f_res ? f_rs : 0);
    uint2 epv[PRE ? 2 : 1][2];
    if constexpr (PRE) if (!f_plain && (M < 0 || f_bwd || f_res)) {      // (PRE = false -- the two-stage image, whose registers hold two weight blocks: requested where it is used)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const long long er = min(t * kFgRows + rt * 16 + r, a.rows - 1) >> ep_sh;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) epv[rt][nt] = *reinterpret_cast<const uint2*>(ep + (size_t)er * ep_ld + wave * 32 + nt * 16 + kg * 4);
        }
    }
    f32x4 acc[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[rt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8 xf[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) xf[rt] = *reinterpret_cast<const bf16x8*>(xs + (rt * 16 + r) * kFgXPitch + (ks * 32 + kg * 8) * 2);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[rt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][nt], xf[rt], acc[rt][nt], 0, 0, 0);
#ifdef VLG_FG_ONE_KSTEP   // tools/ ablation (results are wrong): one of the eight contraction steps
        break;
#endif
    }
    const long long row0 = t * kFgRows;
    constexpr bool kPlain = KS > 8;     // the K = 512 image: a plain layer's adjoint (no mask, no draw, no group sum: 128 of its 256 registers hold the weights)
    float km[2][2][4];
    if (!kPlain && f_rng) {       // (uniform) the dropout draws of the tile, indexed by the OUTPUT row forward and by the row backward (as vlg_ff_act*)
        const long long ra = row0 + r, rb = row0 + 16 + r;
        const bool by_row = f_bwd || f_plain;
        const long long ka = by_row ? ra : (ra >> f_rs) * a.om + (long long)y * a.oy + (ra & ((1 << f_rs) - 1));
        const long long kb = by_row ? rb : (rb >> f_rs) * a.om + (long long)y * a.oy + (rb & ((1 << f_rs) - 1));
        fg_tile_keep(a, ka, kb, wave, kg, f_plain ? a.ncols >> 3 : kFgH >> 3, f_plain ? y * kFgH : 0, km);
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const long long row = row0 + rt * 16 + r;
        const bool live = row < a.rows;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = wave * 32 + nt * 16 + kg * 4;        // this lane's four channels
            float val[4] = {acc[rt][nt][0], acc[rt][nt][1], acc[rt][nt][2], acc[rt][nt][3]};
            if (!kPlain && !f_bwd) {
                // (the pair this replaces rounds the Linear's output to bf16 before the skip connection is added: the same rounding here, so
                //  that the two paths agree to the last bit of what the next layer reads wherever the product's own summation order does)
#pragma unroll
                for (int k = 0; k < 4; ++k) val[k] = fg_round(val[k] + bias4[nt][k]);
                if (f_plain) {        // (uniform) the product alone, or times the counter-based draw over out's [rows, ncols] elements (as vlg_dropout)
                    if (f_rng) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] *= km[rt][nt][k];
                    }
                    if (live && y * kFgH + c < a.ncols) fg_store4(a.out + (size_t)row * a.ldo + y * kFgH + c, val);
                    continue;
                }
                const long long m = row >> f_rs;
                const long long orow = m * a.om + (long long)y * a.oy + (row & ((1 << f_rs) - 1));
                if (live) {
                    if (f_res) {
                        float t4[4];
                        if constexpr (PRE) unpack4(epv[rt][nt], t4);
                        else load4(a.res + (size_t)m * kFgH + c, t4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] += t4[k];
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) val[k] = leaky(val[k], a.slope);
                    if (f_mask) {
                        float t4[4];
                        load4(a.mask + (size_t)orow * kFgH + c, t4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] *= t4[k] * a.mask_scale;
                    } else if (f_rng) {      // the draw of the 8-channel group this lane holds half of (the element indexing of vlg_ff_act)
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] *= km[rt][nt][k];
                    }
#ifndef VLG_FG_NOSTORE     // tools/ ablation
                    fg_store4(a.out + (size_t)orow * kFgH + c, val);
                    if (ys) fg_store4(reinterpret_cast<uint16_t*>(ys + (rt * 16 + r) * kFgXPitch) + c, val);
#else
                    if (val[0] == 123.456f) fg_store4(a.out + (size_t)orow * kFgH + c, val);
#endif
                }
            } else {
                const int J = f_J, j = (int)(row & (J - 1));
                const long long m = row >> f_lj, orow = f_swap ? m * 4 + fg_swap2(j) : row;      // (J = 1 << lj: no 64-bit division per element)
#pragma unroll
                for (int k = 0; k < 4; ++k) val[k] = fg_round(val[k]);      // (the product's bf16 output, as the pair rounds it)
                float sv[4] = {0.f, 0.f, 0.f, 0.f};
                if (live) {
                    float av[4];
                    if constexpr (PRE) unpack4(epv[rt][nt], av);
                    else load4(a.act + (size_t)row * kFgH + c, av);
                    if (!kPlain && f_mask) {
                        float t4[4];
                        load4(a.mask + (size_t)row * kFgH + c, t4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] *= t4[k] * a.mask_scale;
                    } else if (!kPlain && f_rng) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) val[k] *= km[rt][nt][k];
                    }
                    if (kPlain && f_add) {      // (uniform)
                        const float4 o = *reinterpret_cast<const float4*>(a.add + (size_t)row * kFgH + c);
                        val[0] += o.x; val[1] += o.y; val[2] += o.z; val[3] += o.w;
                        if (row < a.M0) {
                            if (a.drop_head) {
                                const float4 mk = *reinterpret_cast<const float4*>(a.drop_head + (size_t)(row / a.L) * kFgH + c);
                                val[0] *= mk.x; val[1] *= mk.y; val[2] *= mk.z; val[3] *= mk.w;
                            }
                        } else if (a.drop_small) {
                            const float mk = a.drop_small[row - a.M0];
#pragma unroll
                            for (int k = 0; k < 4; ++k) val[k] *= mk;
                        }
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        val[k] = av[k] > 0.f ? val[k] : val[k] * a.slope;
                        sv[k] = fg_round(val[k]);                           // the sum of what the next product reads
                    }
                    fg_store4(a.out + (size_t)orow * kFgH + c, val);
                    if (ys) fg_store4(reinterpret_cast<uint16_t*>(ys + (rt * 16 + r) * kFgXPitch) + c, val);
                }
                if (!kPlain && f_sum) {     // (uniform) the J rows of a group sit on J consecutive lanes: the group's sum in row order j = 0, 1, ..
                    float s[4];
                    if (J == 1) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) s[k] = sv[k];
                    } else {
                        // fixed order ((j0 + j1) + (j2 + j3)): every lane of the group computes the same sum
#pragma unroll
                        for (int k = 0; k < 4; ++k) s[k] = fg_group_sum(sv[k], J);
                    }
                    if (live && j == 0) {
                        float* sp = a.sum + (size_t)m * kFgH + c;
                        if (f_acc) {
                            const float4 o = *reinterpret_cast<const float4*>(sp);
                            s[0] += o.x; s[1] += o.y; s[2] += o.z; s[3] += o.w;
                        }
                        *reinterpret_cast<float4*>(sp) = make_float4(s[0], s[1], s[2], s[3]);
                    }
                }
            }
        }
    }
}

// TWO: a second 256 -> 256 stage `b` runs on the first one's stored rows without their round trip through memory: stage a's epilogue writes its
// tile into a third LDS image as well, and stage b's product reads its input fragments there (both weight blocks in registers: 128 VGPRs).  Stage a
// keeps its rows (one column block, no row permutation); it still writes its own output -- the weight gradients read it.
template <int KS, bool TWO, int MA, int MB>   // K = 32 KS input channels; MA / MB: the stages' option words (-1: read from the argument blocks)
__device__ __forceinline__ void ff_gemm_act_body(const FgArgs& a, const FgArgs& b) {
    constexpr int K = KS * 32, kFgXPitch = fg_pitch(K), SEGS = K / 8, SPT = (kFgRows * SEGS + kFgThreads - 1) / kFgThreads;   // 16-byte segments per row / per thread and tile
    static_assert(SPT >= 1 && SPT <= 4, "segments per thread");
    static_assert(!TWO || KS == 8, "the second stage reads a 256-channel image");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const xs0 = smem;
    char* const xs1 = smem + kFgRows * kFgXPitch;
    char* const ys = TWO ? smem + 2 * kFgRows * kFgXPitch : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int y = blockIdx.y;
    const long long tiles = (a.rows + kFgRows - 1) / kFgRows, G = gridDim.x;
    long long t = blockIdx.x;
    if (t >= tiles) return;
    // Input tile staging: thread t moves the 16-byte segments v = t + 512 j (j < SPT) of a tile (row v / SEGS, segment v % SEGS) through named
    // registers (as an array, written under a condition and read an epilogue later, the segments were kept in scratch memory).  TWO tiles are in
    // flight -- set A holds the tile after this one, set B the one after that.
    uint4 xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
    xa1 = xa2 = xa3 = xb1 = xb2 = xb3 = make_uint4(0, 0, 0, 0);
#define FG_V(J_) min(tid + kFgThreads * (J_), kFgRows * SEGS - 1)
#define FG_SEG_LOAD(T_, J_) \
    *reinterpret_cast<const uint4*>(a.x + (size_t)min((T_) * kFgRows + FG_V(J_) / SEGS, a.rows - 1) * a.ldx + (FG_V(J_) % SEGS) * 8)
#define FG_SEG_STORE(XS_, J_, V_) \
    if (tid + kFgThreads * (J_) < kFgRows * SEGS) *reinterpret_cast<uint4*>((XS_) + (FG_V(J_) / SEGS) * kFgXPitch + (FG_V(J_) % SEGS) * 16) = (V_)
#define FG_LOAD_A(T_) { xa0 = FG_SEG_LOAD(T_, 0); if (SPT > 1) xa1 = FG_SEG_LOAD(T_, 1); if (SPT > 2) { xa2 = FG_SEG_LOAD(T_, 2); xa3 = FG_SEG_LOAD(T_, 3); } }
#define FG_LOAD_B(T_) { xb0 = FG_SEG_LOAD(T_, 0); if (SPT > 1) xb1 = FG_SEG_LOAD(T_, 1); if (SPT > 2) { xb2 = FG_SEG_LOAD(T_, 2); xb3 = FG_SEG_LOAD(T_, 3); } }
#define FG_STORE_A(XS_) { FG_SEG_STORE(XS_, 0, xa0); if (SPT > 1) { FG_SEG_STORE(XS_, 1, xa1); } if (SPT > 2) { FG_SEG_STORE(XS_, 2, xa2); FG_SEG_STORE(XS_, 3, xa3); } }
#define FG_STORE_B(XS_) { FG_SEG_STORE(XS_, 0, xb0); if (SPT > 1) { FG_SEG_STORE(XS_, 1, xb1); } if (SPT > 2) { FG_SEG_STORE(XS_, 2, xb2); FG_SEG_STORE(XS_, 3, xb3); } }
    FG_LOAD_A(t)
    // ---- the weight block of this workgroup's column block, in registers for the whole launch ----
    bf16x8 wf[KS][2];
    if (KS > 8 || !a.w_kn) {
        // (K = 512: two [256][256] blocks one after the other, contraction channels 0..255 | 256..511 -- what vlg_ff_transpose256 writes for the
        //  two halves of a [512, 256] weight)
        constexpr int KP = K < kFgH ? K : kFgH;
        const uint16_t* wb = a.w + ((size_t)y * kFgH + wave * 32) * KP;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#ifndef VLG_FG_NOW         // tools/ ablation: every fragment from the first 16 rows of the block
                wf[ks][nt] = *reinterpret_cast<const bf16x8*>(wb + (size_t)(ks >> 3) * kFgH * kFgH + (size_t)(nt * 16 + r) * KP + (ks & 7) * 32 + kg * 8);
#else
                wf[ks][nt] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)r * KP + kg * 8);
#endif
    } else {      // w [K][ldw], w[k][n] (the eight elements of a fragment are eight rows of w apart: 2-byte reads, once per launch)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = y * kFgH + wave * 32 + nt * 16 + r;
            const bool ok = n < a.ncols;
            const uint16_t* wc = a.w + (ok ? n : 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                typedef short fg_v8 __attribute__((ext_vector_type(8)));
                fg_v8 v;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = ok ? (short)wc[(size_t)(ks * 32 + kg * 8 + i) * a.ldw] : (short)0;
                wf[ks][nt] = __builtin_bit_cast(bf16x8, v);
            }
        }
    }
    float bias4[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 4; ++k) bias4[nt][k] = a.bias ? bf2f(a.bias[y * kFgH + wave * 32 + nt * 16 + kg * 4 + k]) : 0.f;
    // ---- the second stage's weight block and bias ----
    bf16x8 wf2[TWO ? KS : 1][2];
    float bias2[2][4];
    if constexpr (TWO) {
        const uint16_t* wb = b.w + (size_t)(wave * 32) * kFgH;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wf2[ks][nt] = *reinterpret_cast<const bf16x8*>(wb + (size_t)(nt * 16 + r) * kFgH + ks * 32 + kg * 8);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int k = 0; k < 4; ++k) bias2[nt][k] = b.bias ? bf2f(b.bias[wave * 32 + nt * 16 + kg * 4 + k]) : 0.f;
    }
    auto tile = [&](long long tt, const char* xs) {
        if constexpr (TWO) {
            fg_tile<KS, false, MA>(a, tt, xs, wf, bias4, tid, y, ys);
            __syncthreads();                                   // stage a's rows are in `ys` (the next write of it is behind the trip's barrier)
            fg_tile<KS, false, MB>(b, tt, ys, wf2, bias2, tid, 0);
        } else {
            fg_tile<KS, true, MA>(a, tt, xs, wf, bias4, tid, y);
        }
    };
    FG_STORE_A(xs0)
    // (tiles past the end re-read the last row: no branches around the loads; their registers are never stored)
    FG_LOAD_A(t + G)
    if constexpr (TWO) {       // ONE tile in flight (a tile is two stages long here, and the second set's registers hold weights)
        __syncthreads();
        for (;;) {
            tile(t, xs0);
            if (t + G >= tiles) break;
            FG_STORE_A(xs1)
            FG_LOAD_A(t + 2 * G)
            __syncthreads();
            tile(t + G, xs1);
            if (t + 2 * G >= tiles) break;
            FG_STORE_A(xs0)
            FG_LOAD_A(t + 3 * G)
            __syncthreads();
            t += 2 * G;
        }
        return;
    }
    FG_LOAD_B(t + 2 * G)
    __syncthreads();
    for (;;) {
        tile(t, xs0);                                          // tile t from image 0
        if (t + G >= tiles) break;
        FG_STORE_A(xs1)                                        // tile t + G -> image 1 (only set A's reads are waited for)
        FG_LOAD_A(t + 3 * G)
        __syncthreads();
        tile(t + G, xs1);                                      // tile t + G from image 1
        if (t + 2 * G >= tiles) break;
        FG_STORE_B(xs0)
        FG_LOAD_B(t + 4 * G)
        __syncthreads();
        t += 2 * G;
    }
}

template <int KS, int M>
__global__ __launch_bounds__(kFgThreads) void ff_gemm_act_kernel(const FgArgs a) { ff_gemm_act_body<KS, false, M, -1>(a, a); }

template <int MA, int MB>
__global__ __launch_bounds__(kFgThreads) void ff_gemm_act2_kernel(const FgArgs a, const FgArgs b) { ff_gemm_act_body<8, true, MA, MB>(a, b); }

// out[z][n][k] = w_z[k][n] for up to eight 256 x 256 bf16 matrices (the layers' weights as the backward launches read them): 32 x 32 tiles through LDS
struct FgTr { const uint16_t* w[8]; };
__global__ __launch_bounds__(256) void ff_transpose256_kernel(const FgTr a, uint16_t* __restrict__ out) {
    __shared__ uint16_t tile[32][34];
    const uint16_t* w = a.w[blockIdx.z];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5, k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + 8 * i][tx] = w[(size_t)(k0 + ty + 8 * i) * kFgH + n0 + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) out[((size_t)blockIdx.z * kFgH + n0 + ty + 8 * i) * kFgH + k0 + tx] = tile[tx][ty + 8 * i];
}

#undef FG_V
#undef FG_SEG_LOAD
#undef FG_SEG_STORE
#undef FG_LOAD_A
#undef FG_LOAD_B
#undef FG_STORE_A
#undef FG_STORE_B

// the option word of an argument block (what fg_tile reads from it in the generic image)
int fg_mode(const FgArgs& a) {
    return (a.bwd ? kMBwd : 0) | (a.res ? kMRes : 0) | (a.mask ? kMMask : 0) | (a.rng ? kMRng : 0) | (a.sum ? kMSum : 0) | (a.swap ? kMSwap : 0) |
           (a.accumulate ? kMAcc : 0) | (a.plain ? kMPlain : 0) | (a.J == 2 ? kMJ2 : 0) | (a.J == 4 ? kMJ4 : 0) | (a.rs == 1 ? kMRs1 : 0) | (a.add ? kMAdd : 0);
}

template <int KS, int M>
int fg_launch_km(const FgArgs& a, int nb, hipStream_t s) {
    static bool attr_set = false;
    constexpr int lds = fg_lds(KS * 32);
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ff_gemm_act_kernel<KS, M>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_set = true;
    }
    const long long tiles = (a.rows + kFgRows - 1) / kFgRows;
    // ONE workgroup per CU holds its weight block for the whole launch (the block is re-read from the L2 by every workgroup: with 512
    // workgroups those reads -- 64 MB of 64-byte pieces out of the same 128 KB -- took ~15 us per launch); the tiles are dealt round-robin
    const int per_block = std::max(1, 256 / nb);
    const int gx = (int)std::min<long long>(tiles, per_block);
    hipLaunchKernelGGL((ff_gemm_act_kernel<KS, M>), dim3(gx, nb), dim3(kFgThreads), lds, s, a);
    return check_launch("ff_gemm_act_kernel");
}

// the images with their options compiled in: the combinations vlgae_amd.parser_ff / align.linear_kn launch in a training step (VLG_FF_GENERIC=1: the
// generic image everywhere, A/B timing); any other combination of a C-ABI caller takes the generic image
int fg_launch(const FgArgs& a, int k, int nb, hipStream_t s) {
    const int m = VLG_ENV("VLG_FF_GENERIC") ? -2 : fg_mode(a);
#define FG_CASE(KS_, M_) if (m == (M_)) return fg_launch_km<KS_, (M_)>(a, nb, s)
    if (k == 32) {
        FG_CASE(1, kMBwd);
        return fg_launch_km<1, -1>(a, nb, s);
    }
    if (k == 512) {
        FG_CASE(16, kMBwd);
        FG_CASE(16, kMBwd | kMAdd);
        return fg_launch_km<16, -1>(a, nb, s);
    }
    FG_CASE(8, 0);                                         // a plain layer
    FG_CASE(8, kMRes);                                     // (no | has) bottlenecks + skip connection
    FG_CASE(8, kMRes | kMRs1);                             // (left | right) bottlenecks + skip connection
    FG_CASE(8, kMRng);                                     // direction stage + nn.Dropout
    FG_CASE(8, kMBwd);
    FG_CASE(8, kMBwd | kMRng);
    FG_CASE(8, kMBwd | kMJ4 | kMSwap | kMSum);
    FG_CASE(8, kMBwd | kMJ2 | kMSum | kMAcc);
    FG_CASE(8, kMPlain);                                   // linear_kn
    FG_CASE(8, kMPlain | kMRng);
#undef FG_CASE
    return fg_launch_km<8, -1>(a, nb, s);
}

int fg_check(const char* what, const void* x, int ldx, int k, const void* w, long long rows, const void* out) {
    if (rows < 0) return set_error(VLG_ERR_SHAPE, "%s: rows=%lld", what, rows);
    if (k != 32 && k != 256 && k != 512) return set_error(VLG_ERR_SHAPE, "%s: %d input channels (32, 256 or 512)", what, k);
    if (ldx < k || ldx % 8) return set_error(VLG_ERR_SHAPE, "%s: input row stride %d (>= %d, a multiple of 8 elements)", what, ldx, k);
    if (rows > 0 && (!x || !w || !out)) return set_error(VLG_ERR_ARG, "%s: null buffer", what);
    if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) & 15) return set_error(VLG_ERR_ARG, "%s: buffers must be 16-byte aligned", what);
    return 0;
}

}  // namespace

}  // namespace vlg

extern "C" {

int vlg_ff_linear_act(const void* x, int ldx, const void* w, const void* bias, long long rows, int nb, const void* residual, int rs, int om,
                      int oy, const void* mask, float mask_scale, const uint64_t* rng, unsigned site, float p, void* out, float slope,
                      void* stream) {
    using namespace vlg;
    if (int rc = fg_check("ff_linear_act", x, ldx, 256, w, rows, out)) return rc;
    if (nb < 1 || nb > 2 || rs < 0 || rs > 1 || om < 1 || oy < 0) return set_error(VLG_ERR_ARG, "ff_linear_act: nb=%d rs=%d om=%d oy=%d", nb, rs, om, oy);
    if (mask && rng) return set_error(VLG_ERR_ARG, "ff_linear_act: mask and rng are exclusive");
    if (rows == 0) return 0;
    FgArgs a{};
    a.x = (const uint16_t*)x; a.ldx = ldx; a.w = (const uint16_t*)w; a.bias = (const uint16_t*)bias; a.rows = rows; a.slope = slope;
    a.res = (const uint16_t*)residual; a.rs = rs; a.om = om; a.oy = oy;
    a.bwd = 0; a.J = 1;
    a.mask = (const uint16_t*)mask; a.mask_scale = rng ? drop_scale(p) : mask_scale; a.rng = rng; a.site = site; a.thr = rng ? drop_threshold(p) : 0;
    a.out = (uint16_t*)out;
    return fg_launch(a, 256, nb, (hipStream_t)stream);
}

int vlg_ff_linear_act_backward(const void* g, int ldg, const void* w_t, int k, int w_kn, long long rows, int J, const void* act, const void* mask, float mask_scale,
                               const uint64_t* rng, unsigned site, float p, void* out, float* sum, int swap, int accumulate, float slope,
                               void* stream) {
    using namespace vlg;
    if (int rc = fg_check("ff_linear_act_backward", g, ldg, k, w_t, rows, out)) return rc;
    if ((w_kn != 0) != (k == 32)) return set_error(VLG_ERR_ARG, "ff_linear_act_backward: the [k][256] weight layout (w_kn) is the one of k = 32 and only its");
    if (k == 512 && (J != 1 || mask || rng || sum)) return set_error(VLG_ERR_ARG, "ff_linear_act_backward: k = 512 is a plain layer's adjoint (J = 1, no mask, draw or sum)");
    if ((J != 1 && J != 2 && J != 4) || rows % J || (swap && J != 4)) return set_error(VLG_ERR_ARG, "ff_linear_act_backward: rows=%lld J=%d swap=%d", rows, J, swap);
    if ((mask || rng) && swap) return set_error(VLG_ERR_ARG, "ff_linear_act_backward: no permutation with a mask");
    if (mask && rng) return set_error(VLG_ERR_ARG, "ff_linear_act_backward: mask and rng are exclusive");
    if (rows > 0 && !act) return set_error(VLG_ERR_ARG, "ff_linear_act_backward: null activation");
    if (rows == 0) return 0;
    FgArgs a{};
    a.x = (const uint16_t*)g; a.ldx = ldg; a.w = (const uint16_t*)w_t; a.w_kn = w_kn; a.ldw = kFgH; a.ncols = kFgH; a.bias = nullptr; a.rows = rows; a.slope = slope;
    a.bwd = 1; a.J = J; a.lj = J == 4 ? 2 : (J == 2 ? 1 : 0); a.swap = swap; a.accumulate = accumulate; a.act = (const uint16_t*)act; a.sum = sum;
    a.mask = (const uint16_t*)mask; a.mask_scale = rng ? drop_scale(p) : mask_scale; a.rng = rng; a.site = site; a.thr = rng ? drop_threshold(p) : 0;
    a.out = (uint16_t*)out;
    return fg_launch(a, k, 1, (hipStream_t)stream);
}

// one stage of vlg_ff_linear_act_chain2 -> the kernel's argument block (x / rows filled by the caller)
static int fg_stage(const char* what, const VlgFfStage& q, int backward, bool first, float slope, vlg::FgArgs& a) {
    using namespace vlg;
    if (!q.w || !q.out) return set_error(VLG_ERR_ARG, "%s: null weight / output", what);
    if (((uintptr_t)q.w | (uintptr_t)q.out | (uintptr_t)q.act | (uintptr_t)q.mask | (uintptr_t)q.sum) & 15) return set_error(VLG_ERR_ARG, "%s: buffers must be 16-byte aligned", what);
    if (q.mask && q.rng) return set_error(VLG_ERR_ARG, "%s: mask and rng are exclusive", what);
    a = FgArgs{};
    a.w = (const uint16_t*)q.w; a.slope = slope; a.out = (uint16_t*)q.out; a.om = 1; a.J = 1;
    a.mask = (const uint16_t*)q.mask; a.mask_scale = q.rng ? drop_scale(q.p) : q.mask_scale; a.rng = q.rng; a.site = q.site; a.thr = q.rng ? drop_threshold(q.p) : 0;
    if (!backward) {
        a.bias = (const uint16_t*)q.bias;
        return 0;
    }
    if (!q.act) return set_error(VLG_ERR_ARG, "%s: null activation", what);
    const int J = q.J ? q.J : 1;
    if ((J != 1 && J != 2 && J != 4) || (q.swap && J != 4) || ((q.mask || q.rng) && q.swap)) return set_error(VLG_ERR_ARG, "%s: J=%d swap=%d", what, J, q.swap);
    if (first && (J != 1 || q.sum || q.swap)) return set_error(VLG_ERR_ARG, "%s: the first stage keeps its rows (J = 1, no sum, no permutation)", what);
    a.bwd = 1; a.J = J; a.lj = J == 4 ? 2 : (J == 2 ? 1 : 0); a.swap = q.swap; a.accumulate = q.accumulate; a.act = (const uint16_t*)q.act; a.sum = q.sum;
    return 0;
}

int vlg_ff_linear_act_chain2(const void* x, int ldx, long long rows, int backward, const VlgFfStage* s1, const VlgFfStage* s2, float slope, void* stream) {
    using namespace vlg;
    if (!s1 || !s2) return set_error(VLG_ERR_ARG, "ff_linear_act_chain2: null stage");
    if (int rc = fg_check("ff_linear_act_chain2", x, ldx, 256, s1->w, rows, s1->out)) return rc;
    FgArgs a, b;
    if (int rc = fg_stage("ff_linear_act_chain2 (stage 1)", *s1, backward, true, slope, a)) return rc;
    if (int rc = fg_stage("ff_linear_act_chain2 (stage 2)", *s2, backward, false, slope, b)) return rc;
    if (backward && rows % b.J) return set_error(VLG_ERR_ARG, "ff_linear_act_chain2: rows=%lld J=%d", rows, b.J);
    if (rows == 0) return 0;
    a.x = (const uint16_t*)x; a.ldx = ldx; a.rows = rows;
    b.x = (const uint16_t*)s1->out; b.ldx = kFgH; b.rows = rows;          // (the two-stage image reads it from LDS; the two-launch route from memory)
    constexpr int lds = 3 * kFgRows * fg_pitch(256);        // 51 KB: below the 64 KB that need no attribute
    static_assert(lds <= 64 * 1024, "dynamic LDS attribute needed");
    const long long tiles = (rows + kFgRows - 1) / kFgRows;
    const dim3 grid((unsigned)std::min<long long>(tiles, 256));
    const int ma = VLG_ENV("VLG_FF_GENERIC") ? -2 : fg_mode(a), mb = fg_mode(b);
    if (ma == kMRng && mb == 0) hipLaunchKernelGGL((ff_gemm_act2_kernel<kMRng, 0>), grid, dim3(kFgThreads), lds, (hipStream_t)stream, a, b);
    else if (ma == (kMBwd | kMRng) && mb == (kMBwd | kMJ4 | kMSwap | kMSum))
        hipLaunchKernelGGL((ff_gemm_act2_kernel<kMBwd | kMRng, kMBwd | kMJ4 | kMSwap | kMSum>), grid, dim3(kFgThreads), lds, (hipStream_t)stream, a, b);
    else {   // any other pair of option words: the two stages as two launches (the same bits; a generic two-stage image would not fit 256 registers)
        if (int rc = fg_launch(a, 256, 1, (hipStream_t)stream)) return rc;
        return fg_launch(b, 256, 1, (hipStream_t)stream);
    }
    return check_launch("ff_gemm_act2_kernel");
}

int vlg_ff_linear_kn(const void* x, int ldx, const void* w, int ldw, long long rows, int ncols, const uint64_t* rng, unsigned site, float p, void* out, int ldo,
                     void* stream) {
    using namespace vlg;
    if (int rc = fg_check("ff_linear_kn", x, ldx, 256, w, rows, out)) return rc;
    if (ncols < 8 || ncols % 8 || ldw < ncols || ldo < ncols || ldo % 4 || ncols > 64 * kFgH)
        return set_error(VLG_ERR_SHAPE, "ff_linear_kn: ncols=%d (a multiple of 8, <= 16384) ldw=%d ldo=%d (>= ncols, ldo a multiple of 4)", ncols, ldw, ldo);
    if (rows == 0) return 0;
    FgArgs a{};
    a.x = (const uint16_t*)x; a.ldx = ldx; a.w = (const uint16_t*)w; a.w_kn = 1; a.ldw = ldw; a.ncols = ncols; a.plain = 1; a.ldo = ldo;
    a.rows = rows; a.J = 1; a.out = (uint16_t*)out;
    a.rng = rng; a.site = site; a.mask_scale = rng ? drop_scale(p) : 1.f; a.thr = rng ? drop_threshold(p) : 0;
    return fg_launch(a, 256, (ncols + kFgH - 1) / kFgH, (hipStream_t)stream);
}

int vlg_ff_linear_mlp_act_backward(const void* g, int ldg, const void* w_t, long long rows, const float* add, const void* x, const float* drop_head,
                                   const float* drop_small, long long M0, int L, void* out, float slope, void* stream) {
    using namespace vlg;
    if (int rc = fg_check("ff_linear_mlp_act_backward", g, ldg, 512, w_t, rows, out)) return rc;
    if (M0 < 0 || M0 > rows || L < 1 || M0 % L || M0 > 0x7fffffffLL) return set_error(VLG_ERR_SHAPE, "ff_linear_mlp_act_backward: rows=%lld M0=%lld L=%d", rows, M0, L);
    if (rows > 0 && (!add || !x)) return set_error(VLG_ERR_ARG, "ff_linear_mlp_act_backward: null buffer");
    if (((uintptr_t)add | (uintptr_t)x | (uintptr_t)drop_head) & 15) return set_error(VLG_ERR_ARG, "ff_linear_mlp_act_backward: buffers must be 16-byte aligned");
    if (rows == 0) return 0;
    FgArgs a{};
    a.x = (const uint16_t*)g; a.ldx = ldg; a.w = (const uint16_t*)w_t; a.rows = rows; a.slope = slope;
    a.bwd = 1; a.J = 1; a.act = (const uint16_t*)x; a.out = (uint16_t*)out;
    a.add = add; a.drop_head = drop_head; a.drop_small = drop_small; a.M0 = (int)M0; a.L = L;
    return fg_launch(a, 512, 1, (hipStream_t)stream);
}

int vlg_ff_transpose256(const void* const* w, int n, void* out, void* stream) {
    using namespace vlg;
    if (n < 1 || n > 8 || !out || !w) return set_error(VLG_ERR_ARG, "ff_transpose256: n=%d (1..8) or a null buffer", n);
    FgTr a{};
    for (int z = 0; z < n; ++z) {
        if (!w[z]) return set_error(VLG_ERR_ARG, "ff_transpose256: matrix %d is null", z);
        a.w[z] = (const uint16_t*)w[z];
    }
    hipLaunchKernelGGL(ff_transpose256_kernel, dim3(8, 8, n), dim3(256), 0, (hipStream_t)stream, a, (uint16_t*)out);
    return check_launch("ff_transpose256_kernel");
}

}  // extern "C"
