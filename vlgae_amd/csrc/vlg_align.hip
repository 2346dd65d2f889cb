// vlg_align.hip -- region x word alignment kernels (gfx950) and their C-ABI entry points.
//
//   vlg_bilinear_align : DependencyBoxRel.gather_logit_simple  (src/model/joint.py:406-419)
//   (the attention-fuse feeding the parser lives in vlg_attn.hip)
//
// The contraction runs on the matrix cores: v_mfma_f32_16x16x32_bf16 for bf16 features (fp32
// accumulate), v_mfma_f32_16x16x4_f32 for fp32 features (exact fp32 products, the reference's numerics).
// Both operands are K-contiguous in memory ([.,.,d]), which IS the MFMA fragment order, so fragments
// are loaded straight from global memory with one 16-byte load per lane -- no LDS staging, no
// transposes.  The epilogue (mask -> -INF, max over V, max over Q, batch diagonal) is fused behind
// the contraction through a per-wave LDS tile, so the [B,A,Q,V] tensor is only written when asked for.
// A generic fp32-FMA kernel remains for shapes the MFMA path does not take (d > 128 or unaligned d).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "vlg_common.h"
#include "vlg_dp_core.h"   // F32In / BF16In element loaders
#include "vlg_ground.h"
#include "vlg_mfma.h"

namespace vlg {

constexpr int kAlignThreads = 256;
constexpr int kQT = 64;   // query rows per tile
constexpr int kVT = 64;   // region rows per tile

__device__ __forceinline__ float neg_infinity() { return __uint_as_float(0xff800000u); }

// One block = one caption b and a contiguous chunk of images a.  Loops q-tiles x v-tiles; the tile
// product lives in LDS so that the full-tensor rows are written coalesced and the two max
// reductions read it without touching HBM again.
template <typename In>
__global__ __launch_bounds__(kAlignThreads) void align_kernel(
    const typename In::T* __restrict__ txt, const typename In::T* __restrict__ vis,
    const uint8_t* __restrict__ tmask, const uint8_t* __restrict__ vmask, int B, int A, int Q, int V, int d,
    float neg_inf, float* __restrict__ out_full, float* __restrict__ out_maxV, float* __restrict__ out_maxQ,
    float* __restrict__ out_diag, int a_per_block) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, b = blockIdx.y;
    const int a0 = blockIdx.x * a_per_block, a1 = min(A, a0 + a_per_block);
    const int ds = d + 1;   // +1 pad: lanes walking different rows hit different banks
    float* txt_s = smem;                       // [kQT][ds]
    float* vis_s = txt_s + kQT * ds;           // [kVT][ds]
    float* res_s = vis_s + kVT * ds;           // [kQT][kVT + 1]
    float* maxV_s = res_s + kQT * (kVT + 1);   // [Q]
    float* maxQ_s = maxV_s + Q;                // [V]
    const int rs = kVT + 1;
    const bool single_q_tile = Q <= kQT;

    for (int a = a0; a < a1; ++a) {
        for (int i = tid; i < Q; i += kAlignThreads) maxV_s[i] = neg_infinity();
        for (int i = tid; i < V; i += kAlignThreads) maxQ_s[i] = neg_infinity();
        for (int q0 = 0; q0 < Q; q0 += kQT) {
            const int qn = min(kQT, Q - q0);
            if (!(single_q_tile && a != a0)) {   // the caption tile is reused across the image chunk
                __syncthreads();
                for (int i = tid; i < qn * d; i += kAlignThreads) {
                    const int q = i / d, k = i - q * d;
                    txt_s[q * ds + k] = In::ld(txt, ((size_t)b * Q + q0 + q) * d + k);
                }
            }
            for (int v0 = 0; v0 < V; v0 += kVT) {
                const int vn = min(kVT, V - v0);
                __syncthreads();
                for (int i = tid; i < vn * d; i += kAlignThreads) {
                    const int v = i / d, k = i - v * d;
                    vis_s[v * ds + k] = In::ld(vis, ((size_t)a * V + v0 + v) * d + k);
                }
                __syncthreads();
                for (int i = tid; i < qn * vn; i += kAlignThreads) {
                    const int q = i / vn, v = i - q * vn;
                    const float* x = txt_s + q * ds;
                    const float* y = vis_s + v * ds;
                    float acc = 0.f;
                    for (int k = 0; k < d; ++k) acc = fmaf(x[k], y[k], acc);
                    const bool keep = (!tmask || tmask[(size_t)b * Q + q0 + q]) && (!vmask || vmask[(size_t)a * V + v0 + v]);
                    res_s[q * rs + v] = keep ? acc : neg_inf;   // joint.py:417-418
                }
                __syncthreads();
                if (out_full)
                    for (int i = tid; i < qn * vn; i += kAlignThreads) {
                        const int q = i / vn, v = i - q * vn;
                        out_full[(((size_t)b * A + a) * Q + q0 + q) * V + v0 + v] = res_s[q * rs + v];
                    }
                if (out_diag && a == b)
                    for (int i = tid; i < qn * vn; i += kAlignThreads) {
                        const int q = i / vn, v = i - q * vn;
                        out_diag[((size_t)b * Q + q0 + q) * V + v0 + v] = res_s[q * rs + v];
                    }
                if (out_maxV)
                    for (int q = tid; q < qn; q += kAlignThreads) {
                        float m = maxV_s[q0 + q];
                        for (int v = 0; v < vn; ++v) m = fmaxf(m, res_s[q * rs + v]);
                        maxV_s[q0 + q] = m;
                    }
                if (out_maxQ)
                    for (int v = tid; v < vn; v += kAlignThreads) {
                        float m = maxQ_s[v0 + v];
                        for (int q = 0; q < qn; ++q) m = fmaxf(m, res_s[q * rs + v]);
                        maxQ_s[v0 + v] = m;
                    }
            }
        }
        __syncthreads();
        if (out_maxV)
            for (int i = tid; i < Q; i += kAlignThreads) out_maxV[((size_t)b * A + a) * Q + i] = maxV_s[i];
        if (out_maxQ)
            for (int i = tid; i < V; i += kAlignThreads) out_maxQ[((size_t)b * A + a) * V + i] = maxQ_s[i];
        __syncthreads();
    }
}

// ---- MFMA path --------------------------------------------------------------------------------------

constexpr int kCTB = 3;      // col tiles (16 regions each) per LDS tile pass: 48 regions
constexpr int kTileVP = kCTB * 16 + 4;   // tile pitch 52: rows stay 16-byte aligned, and 52q + 9p hits 32 distinct banks

// One block = caption b x (4 waves x a_per_wave images); each wave owns its images and its LDS tile.
// d == KCH * KW exactly (dispatch guarantees it), so fragment loads need no K guards: lane l of a 16-row
// operand tile loads elements [row l&15][kc*KW + EPL*(l>>4) .. +EPL) -- one 16-byte load, and the kc offsets
// are instruction immediates.  Rows past the end are clamped; their products are never stored.
// ARGS (the grounding loss's variant, TILE only): also records WHERE each maximum sits (first position on ties, like
// a sequential scan) and, on the diagonal pairs a == b, subtracts the POS prior pen[b,q,seg(v)] before the maxima
// (joint.py:446-470).  Separate instantiations: the plain paths pay nothing for it.
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false));
}

template <int CTRL>
__device__ __forceinline__ int row_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }

struct AlignArgs {
    const float* pen;          // [B,Q,n_seg] or null
    const uint8_t* seg_of_v;   // [V]
    int n_seg;
    uint16_t* argV;            // [B,A,Q] region index of max over V
    uint16_t* argQ;            // [B,A,V] query index of max over Q
};

// DIRECT (fused-maxima dispatch, no full tensor): no LDS tile -- max over V is kept per lane and row across the region tiles
// of an image and folded over the 16 column lanes (DPP) once per image.  The batch-diagonal block, when asked for, comes from
// a second launch over the B diagonal pairs only (a_per_wave < 0).
template <bool F32IN, int KCH, bool TILE, bool ARGS, int RTBV = MfmaCfg<F32IN>::RTB, bool DIRECT = false>
__global__ __launch_bounds__(kAlignThreads) void align_mfma_kernel(
    const typename MfmaCfg<F32IN>::T* __restrict__ txt, const typename MfmaCfg<F32IN>::T* __restrict__ vis,
    const uint8_t* __restrict__ tmask, const uint8_t* __restrict__ vmask, int B, int A, int Q, int V,
    float neg_inf, float* __restrict__ out_full, float* __restrict__ out_maxV, float* __restrict__ out_maxQ,
    float* __restrict__ out_diag, int a_per_wave, AlignArgs xa) {
    static_assert(!ARGS || TILE || DIRECT, "arg-max tracking reads the LDS tile or keeps (max, position) pairs in registers");
    using C = MfmaCfg<F32IN>;
    using Frag = typename C::Frag;
    constexpr int RTB = RTBV, QB = RTB * 16, VB = kCTB * 16, d = KCH * C::KW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    float* tile = reinterpret_cast<float*>(smem_raw) + (size_t)wave * (QB * kTileVP + 2 * QB);   // [QB][kTileVP]
    float* mv = tile + QB * kTileVP;                                                             // [QB] running max over V
    int* mi = reinterpret_cast<int*>(mv + QB);                                                   // [QB] ... and where (ARGS)
    const bool diag_only = a_per_wave < 0;   // only the pairs a == b (the batch-diagonal block by itself): wave 0, image b
    const int a_begin = diag_only ? b : (blockIdx.x * 4 + wave) * a_per_wave;
    const typename C::T* txt_b = txt + (size_t)b * Q * d;
    const int crow = (lane >> 4) * 4, ccol = lane & 15;   // C/D fragment: row = 4*(l>>4) + reg, col = l&15
    const int koff = C::EPL * (lane >> 4);

    for (int q0 = 0; q0 < Q; q0 += QB) {
        const int qn = min(QB, Q - q0);
        Frag afrag[RTB][KCH];   // caption fragments stay in registers across all images of this wave
#pragma unroll
        for (int rt = 0; rt < RTB; ++rt) {
            const Frag* rowp = reinterpret_cast<const Frag*>(txt_b + (size_t)min(q0 + rt * 16 + ccol, Q - 1) * d + koff);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) afrag[rt][kc] = rowp[kc * (C::KW / C::EPL)];
        }
        // query-side keep bits of this lane's C rows (row = 16*rt + 4*(l>>4) + e), hoisted out of the image loop
        unsigned tkeep = 0;
#pragma unroll
        for (int rt = 0; rt < RTB; ++rt)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = q0 + rt * 16 + crow + e;
                const unsigned m = tmask ? tmask[(size_t)b * Q + min(q, Q - 1)] : 1u;
                tkeep |= ((q < Q && m != 0) ? 1u : 0u) << (rt * 4 + e);
            }

        // Work items of this wave in order: (image a, region group v0) -> kCTB region tiles each.  The region
        // fragments (and their keep bits) run through a 3-deep register ring that is always two tiles
        // (~2 x RTB x KCH MFMAs) ahead of the matrix cores, across group and image boundaries.
        const int n_img = diag_only ? 1 : max(0, min(a_per_wave, A - a_begin));
        const int n_grp = (V + VB - 1) / VB, n_items = n_img * n_grp;
        auto load_tile = [&](int item, int ct, Frag* f, unsigned& keep) {
            const int it = min(item, n_items - 1);            // past the end: re-load the last tile, never used
            const int ai = it / n_grp, v0 = (it - ai * n_grp) * VB;
            const int vrow = min(v0 + ct * 16 + ccol, V - 1);
            const size_t arow = (size_t)(a_begin + ai) * V + vrow;
            const Frag* rowp = reinterpret_cast<const Frag*>(vis + arow * d + koff);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) f[kc] = rowp[kc * (C::KW / C::EPL)];
            keep = vmask ? vmask[arow] : 1u;
        };
        Frag f0[KCH], f1[KCH], f2[KCH];
        unsigned k0 = 0, k1 = 0, k2 = 0;
        float rm[DIRECT ? RTB : 1][4];   // DIRECT: running max over V of this lane's C rows, at its column, across an image's groups
        int ri[DIRECT && ARGS ? RTB : 1][4];   // ... and the region that holds it (first one: columns only ascend for a lane)
        // diagonal-only launches produce no maxima, so an image's region groups are independent: they are dealt out over the
        // waves and blocks of the caption instead of running through one wave
        const int i_first = diag_only ? wave + 4 * (int)blockIdx.x : 0, i_step = diag_only ? 4 * (int)gridDim.x : 1;
        if (i_first < n_items) { load_tile(i_first, 0, f0, k0); load_tile(i_first, 1, f1, k1); }

        for (int item = i_first; item < n_items; item += i_step) {
            const int ai = item / n_grp, v0 = (item - ai * n_grp) * VB;
            const int a = a_begin + ai, vn = min(VB, V - v0);
            const size_t ob = ((size_t)b * A + a) * Q;   // row base of this (b, a) pair
            float cmax[kCTB];   // running max over this lane's rows, per region column (for max over Q)
            int cidx[kCTB];     // ... and the row that holds it (ARGS)
            const bool prior_on = ARGS && xa.pen != nullptr && a == b;   // uniform
            const bool use_tile = TILE && (!DIRECT || out_full != nullptr || (out_diag != nullptr && a == b));   // uniform
            if (DIRECT && v0 == 0) {
#pragma unroll
                for (int rt = 0; rt < RTB; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        rm[rt][e] = neg_infinity();
                        if (ARGS) ri[rt][e] = 0x7fff;
                    }
            }

            auto compute = [&](int ct, const Frag* bf, unsigned keepv, auto write_tile) {
                constexpr bool WT = decltype(write_tile)::value;
                // rows past Q and masked rows / columns all take the fill value, so one select + one max per element
                const unsigned lane_keep = (v0 + ct * 16 + ccol < V && keepv != 0) ? tkeep : 0u;
                float cm = neg_infinity();
                float* tcol = tile + crow * kTileVP + ct * 16 + ccol;   // + (16*rt + e) * kTileVP: immediate offsets
                if (ARGS) {
                    // one (max, row) pair per row tile, merged afterwards: a single running pair would chain
                    // 4 RTB dependent compare-selects (the plain path's v_max chain is reassociated by the compiler)
                    const float* prow = nullptr;   // prior row of this lane's first C row, at this column's segment
                    if (prior_on)
                        prow = xa.pen + (size_t)b * Q * xa.n_seg + xa.seg_of_v[min(v0 + ct * 16 + ccol, V - 1)];
                    float pm[RTB];
                    int pi[RTB];
#pragma unroll
                    for (int rt = 0; rt < RTB; ++rt) {
                        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kc = 0; kc < KCH; ++kc) acc = mma_chunk<F32IN>(afrag[rt][kc], bf[kc], acc);
                        float v4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float val = (lane_keep >> (rt * 4 + e)) & 1u ? acc[e] : neg_inf;   // joint.py:417-418
                            if (prior_on) val -= prow[(size_t)min(q0 + rt * 16 + crow + e, Q - 1) * xa.n_seg];   // joint.py:466-469
                            if (WT) tcol[(rt * 16 + e) * kTileVP] = val;
                            if (DIRECT && !WT && val > rm[rt][e]) { rm[rt][e] = val; ri[rt][e] = v0 + ct * 16 + ccol < V ? v0 + ct * 16 + ccol : 0x7fff; }
                            v4[e] = val;
                        }
                        // first maximum of the four rows (ascending): two independent pairs, then merge
                        const bool b01 = v4[1] > v4[0], b23 = v4[3] > v4[2];
                        const float m01 = b01 ? v4[1] : v4[0], m23 = b23 ? v4[3] : v4[2];
                        const int i01 = b01 ? 1 : 0, i23 = b23 ? 3 : 2;
                        const bool bb = m23 > m01;
                        pm[rt] = bb ? m23 : m01;
                        pi[rt] = rt * 16 + crow + (bb ? i23 : i01);
                    }
                    cm = pm[0];
                    int ci = pi[0];
#pragma unroll
                    for (int rt = 1; rt < RTB; ++rt)
                        if (pm[rt] > cm) { cm = pm[rt]; ci = pi[rt]; }   // row tiles ascend: strict > keeps the first maximum
                    cidx[ct] = q0 + ci;
                    return cm;
                }
#pragma unroll
                for (int rt = 0; rt < RTB; ++rt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) acc = mma_chunk<F32IN>(afrag[rt][kc], bf[kc], acc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = (lane_keep >> (rt * 4 + e)) & 1u ? acc[e] : neg_inf;   // joint.py:417-418
                        if (WT) tcol[(rt * 16 + e) * kTileVP] = val;
                        if (DIRECT && !WT) rm[rt][e] = fmaxf(rm[rt][e], val);
                        cm = fmaxf(cm, val);
                    }
                }
                return cm;
            };
            load_tile(item, 2, f2, k2);
            if (use_tile) {
                cmax[0] = compute(0, f0, k0, std::true_type{});
                load_tile(item + i_step, 0, f0, k0);
                cmax[1] = compute(1, f1, k1, std::true_type{});
                load_tile(item + i_step, 1, f1, k1);
                cmax[2] = compute(2, f2, k2, std::true_type{});
            } else {
                cmax[0] = compute(0, f0, k0, std::false_type{});
                load_tile(item + i_step, 0, f0, k0);
                cmax[1] = compute(1, f1, k1, std::false_type{});
                load_tile(item + i_step, 1, f1, k1);
                cmax[2] = compute(2, f2, k2, std::false_type{});
            }
            static_assert(kCTB == 3, "the fragment ring is written for three region tiles per group");

            // ---- max over Q straight from the accumulators: lanes l, l^16, l^32, l^48 hold the same column ----
            if (out_maxQ) {
#pragma unroll
                for (int ct = 0; ct < kCTB; ++ct) {
                    float m = cmax[ct];
                    const int vcol = v0 + ct * 16 + ccol;
                    if (ARGS) {   // (value, row) pairs: larger value wins, equal values keep the smaller row
                        int mi_ = cidx[ct];
#pragma unroll
                        for (int k = 16; k <= 32; k <<= 1) {
                            const float om = __shfl_xor(m, k, 64);
                            const int oi = __shfl_xor(mi_, k, 64);
                            if (om > m || (om == m && oi < mi_)) { m = om; mi_ = oi; }
                        }
                        if (lane < 16 && vcol < V) {
                            const size_t at = ((size_t)b * A + a) * V + vcol;
                            if (q0 == 0 || m > out_maxQ[at]) {   // later row groups only win with a strictly larger value
                                out_maxQ[at] = m;
                                xa.argQ[at] = (uint16_t)mi_;
                            }
                        }
                    } else {
                        m = fmaxf(m, __shfl_xor(m, 16, 64));
                        m = fmaxf(m, __shfl_xor(m, 32, 64));
                        if (lane < 16 && vcol < V) {
                            float* dst = out_maxQ + ((size_t)b * A + a) * V + vcol;
                            *dst = q0 == 0 ? m : fmaxf(*dst, m);   // same wave handles every row group of (b, a)
                        }
                    }
                }
            }
            if (DIRECT && !use_tile && out_maxV && v0 + VB >= V) {
                // ---- max over V of the whole image from the per-lane maxima: fold the 16 column lanes ----
#pragma unroll
                for (int rt = 0; rt < RTB; ++rt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float m = rm[rt][e];   // DPP steps inside the 16-lane row: one v_max each, no LDS round trip
                        const int q = rt * 16 + crow + e;
                        if (ARGS) {   // (value, region) pairs: larger value wins, equal values keep the smaller region
                            int mi_ = ri[rt][e];
                            auto step = [&](float om, int oi) {
                                if (om > m || (om == m && oi < mi_)) { m = om; mi_ = oi; }
                            };
                            step(row_dpp<0xB1>(m), row_dpp_i<0xB1>(mi_));
                            step(row_dpp<0x4E>(m), row_dpp_i<0x4E>(mi_));
                            step(row_dpp<0x141>(m), row_dpp_i<0x141>(mi_));
                            step(row_dpp<0x140>(m), row_dpp_i<0x140>(mi_));
                            if (ccol == 0 && q < qn) {
                                out_maxV[ob + q0 + q] = m;
                                xa.argV[ob + q0 + q] = (uint16_t)mi_;
                            }
                            continue;
                        }
                        m = fmaxf(m, row_dpp<0xB1>(m));    // quad_perm [1,0,3,2]
                        m = fmaxf(m, row_dpp<0x4E>(m));    // quad_perm [2,3,0,1]
                        m = fmaxf(m, row_dpp<0x141>(m));   // row_half_mirror
                        m = fmaxf(m, row_dpp<0x140>(m));   // row_mirror
                        if (ccol == 0 && q < qn) out_maxV[ob + q0 + q] = m;
                    }
            }
            if (TILE && use_tile) {
                __builtin_amdgcn_wave_barrier();
                // ---- copy-out from the wave's LDS tile (pitch kTileVP) to the output block (pitch V) ----
                if (out_full || (out_diag && a == b)) {
                    float* dst_full = out_full ? out_full + (ob + q0) * V + v0 : nullptr;
                    float* dst_diag = (out_diag && a == b) ? out_diag + ((size_t)b * Q + q0) * V + v0 : nullptr;
                    if ((vn & 3) == 0 && (V & 3) == 0) {
                        // 16 bytes per lane: vn/4 lanes per row, 64/(vn/4) rows per step, two steps in flight
                        const int c4 = vn >> 2, rps = 64 / c4;
                        const int lr = lane / c4, lc = lane - lr * c4;
                        const bool act = lr < rps;
                        for (int qb = 0; qb < qn; qb += 2 * rps) {
                            const int qa = qb + lr, qc = qa + rps;
                            float4 ta, tc;
                            if (act && qa < qn) ta = *reinterpret_cast<const float4*>(tile + qa * kTileVP + lc * 4);
                            if (act && qc < qn) tc = *reinterpret_cast<const float4*>(tile + qc * kTileVP + lc * 4);
                            if (act && qa < qn) {
                                if (dst_full) *reinterpret_cast<float4*>(dst_full + (size_t)qa * V + lc * 4) = ta;
                                if (dst_diag) *reinterpret_cast<float4*>(dst_diag + (size_t)qa * V + lc * 4) = ta;
                            }
                            if (act && qc < qn) {
                                if (dst_full) *reinterpret_cast<float4*>(dst_full + (size_t)qc * V + lc * 4) = tc;
                                if (dst_diag) *reinterpret_cast<float4*>(dst_diag + (size_t)qc * V + lc * 4) = tc;
                            }
                        }
                    } else {   // odd widths: scalar, (q, v) advanced with carry instead of a division per element
                        const int dq = 64 / vn, dv = 64 - dq * vn;
                        int q = lane / vn, v = lane - q * vn;
                        for (int idx = lane; idx < qn * vn; idx += 64) {
                            const float val = tile[q * kTileVP + v];
                            if (dst_full) dst_full[(size_t)q * V + v] = val;
                            if (dst_diag) dst_diag[(size_t)q * V + v] = val;
                            q += dq;
                            v += dv;
                            if (v >= vn) { v -= vn; ++q; }
                        }
                    }
                }
                if (out_maxV) {
                    // 4 lanes per query row, each scans a quarter of the regions; 16 rows per pass
                    const int part = lane & 3, vq = (vn + 3) >> 2, vlo = part * vq, vhi = min(vn, vlo + vq);
                    static_assert(kCTB * 16 <= 48, "a quarter row is at most 12 regions");
                    for (int qb = 0; qb < qn; qb += 16) {
                        const int q = qb + (lane >> 2);
                        const float* row = tile + min(q, qn - 1) * kTileVP;
                        float r[12];   // all twelve reads in flight together, then a max tree
#pragma unroll
                        for (int u = 0; u < 12; ++u) r[u] = row[min(vlo + u, kTileVP - 1)];
                        float m0 = neg_infinity(), m1 = neg_infinity();
#pragma unroll
                        for (int u = 0; u < 12; u += 2) {
                            m0 = fmaxf(m0, vlo + u < vhi ? r[u] : neg_infinity());
                            m1 = fmaxf(m1, vlo + u + 1 < vhi ? r[u + 1] : neg_infinity());
                        }
                        float m = fmaxf(m0, m1);
                        if (ARGS) {
                            // first maximum of this lane's quarter (ascending positions): pairwise tree, the
                            // earlier position wins ties at every merge
                            float tm[12];
                            int ti[12];
#pragma unroll
                            for (int u = 0; u < 12; ++u) {
                                tm[u] = vlo + u < vhi ? r[u] : neg_infinity();
                                ti[u] = vlo + u;
                            }
#pragma unroll
                            for (int span = 1; span < 12; span <<= 1)
#pragma unroll
                                for (int u = 0; u + span < 12; u += 2 * span)
                                    if (tm[u + span] > tm[u]) { tm[u] = tm[u + span]; ti[u] = ti[u + span]; }
                            m = tm[0];
                            int mi_ = ti[0];
                            if (vlo >= vhi) mi_ = 0x7fff;   // empty quarter: never wins a tie
#pragma unroll
                            for (int k = 1; k <= 2; k <<= 1) {
                                const float om = __shfl_xor(m, k, 64);
                                const int oi = __shfl_xor(mi_, k, 64);
                                if (om > m || (om == m && oi < mi_)) { m = om; mi_ = oi; }
                            }
                            mi_ += v0;
                            if (part == 0 && q < qn) {
                                if (v0 != 0 && !(m > mv[q])) { m = mv[q]; mi_ = mi[q]; }   // earlier groups keep ties
                                if (v0 + VB >= V) {
                                    out_maxV[ob + q0 + q] = m;
                                    xa.argV[ob + q0 + q] = (uint16_t)mi_;
                                } else {
                                    mv[q] = m;
                                    mi[q] = mi_;
                                }
                            }
                        } else {
                            m = fmaxf(m, __shfl_xor(m, 1, 64));
                            m = fmaxf(m, __shfl_xor(m, 2, 64));
                            if (part == 0 && q < qn) {
                                if (v0 != 0) m = fmaxf(m, mv[q]);
                                if (v0 + VB >= V) out_maxV[ob + q0 + q] = m;
                                else mv[q] = m;
                            }
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

// =====================================================================================================
// Fused maxima, one region group per image (V <= 48), bf16 features, d = 128: max over V and max over Q of every
// (caption, image) block without the [B,A,Q,V] tensor (the grounding loss's / decoder's inputs, joint.py:473-483, :519-524).
//
// One workgroup = 8 captions (one per wavefront) x a range of images.  A wavefront keeps its caption's fragments for
// 96 query rows in registers (the A operand of every MFMA); the image tiles stream through a double-buffered LDS tile
// that all eight wavefronts read, so an image's rows leave L2 once per EIGHT captions (the per-wave streaming of
// align_mfma_kernel moved 1.6 GB L2 -> CU at config-2 and was bound by it).  Tiles are staged through registers
// (global_load_dwordx4 issued one image ahead -> ds_write_b128) into an XOR-swizzled image: 16-byte segment s of
// row r sits at slot 16 r + (s ^ (r & 15)), which makes the MFMA fragment reads (16 rows x one segment per 16-lane
// group) conflict-free without padding.  One barrier per image.
// Epilogue, per 16-row tile, straight from the accumulators: max over the three column tiles element-wise
// (v_max3), then one 16-lane DPP butterfly per register -> row maxima; the running element-wise maximum over the row
// tiles gives the column maxima with one within-lane max and two cross-group shuffles per image.  Masks cost nothing
// when a row tile / an image has none (wave-uniform tests); otherwise one select per element, as the reference's
// masked_fill_ (joint.py:417-418).  Rows / regions past Q / V are clamped duplicates of the last one: they cannot
// change a maximum and are never stored.
// =====================================================================================================
constexpr int kAMThreads = 512, kAMWaves = 8, kAMRows = 48, kAMSlots = kAMRows * 16;   // 16-byte slots per image tile

#define VLG_AM_DPP4(OP, C)                                                                                      \
    asm("s_nop 1\n\t" OP " %0, %0, %0 " C "\n\t" OP " %1, %1, %1 " C "\n\t" OP " %2, %2, %2 " C "\n\t" OP " %3, %3, %3 " C \
        : "+v"(m0), "+v"(m1), "+v"(m2), "+v"(m3))

// ARGS (the grounding loss's variant, joint.py:446-483): also records WHERE each maximum sits (first position on ties, like a
// sequential scan) and, on the diagonal pairs a == b, subtracts the POS prior pen[b,q,seg(v)] before the maxima.  It runs
// 48-row passes (RT = 3): the position registers would not fit next to 96 rows of caption fragments.
template <bool ARGS, int RT>
__global__ __launch_bounds__(kAMThreads, 2) void align_max_kernel(
    const uint16_t* __restrict__ txt, const uint16_t* __restrict__ vis, const uint8_t* __restrict__ tmask,
    const uint8_t* __restrict__ vmask, int B, int A, int Q, int V, float neg_inf, float* __restrict__ out_maxV,
    float* __restrict__ out_maxQ, int a_per_block, AlignArgs xa) {
    constexpr int d = 128, KCH = 4;
    __shared__ uint4 tiles[2][kAMSlots];
    __shared__ uint8_t ckeep_s[2][kAMRows];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y * kAMWaves + wave, bc = min(b, B - 1);   // a wave past the batch mirrors the last caption, stores nothing
    const int a0 = blockIdx.x * a_per_block, n_img = min(A, a0 + a_per_block) - a0;
    const int g = lane >> 4, ccol = lane & 15, crow = g * 4;
    const float ninf = neg_infinity();
    const bool is0 = ccol == 0, is1 = ccol == 1, is2 = ccol == 2;

    // staging: this thread's slots of an image tile (slot -> row, swizzled segment); the second one only for tid < 256
    constexpr int NS = 2;
    const bool has2 = tid < kAMSlots - kAMThreads;
    int soff[NS], srow[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int sl = tid + k * kAMThreads, r = sl >> 4;
        srow[k] = r;
        soff[k] = ((sl & 15) ^ (r & 15)) * 8;
    }
    auto stage_load = [&](int a, uint4* x, unsigned& ck) {
        const uint16_t* img = vis + (size_t)a * V * d;
        x[0] = *reinterpret_cast<const uint4*>(img + (size_t)min(srow[0], V - 1) * d + soff[0]);
        if (has2) x[1] = *reinterpret_cast<const uint4*>(img + (size_t)min(srow[1], V - 1) * d + soff[1]);
        if (vmask && tid < kAMRows) ck = vmask[(size_t)a * V + min(tid, V - 1)];
    };
    auto stage_write = [&](int buf, const uint4* x, unsigned ck) {
        tiles[buf][tid] = x[0];
        if (has2) tiles[buf][tid + kAMThreads] = x[1];
        if (vmask && tid < kAMRows) ckeep_s[buf][tid] = (uint8_t)(ck != 0);
    };
    // fragment read offsets (in 16-byte slots) for K chunk kc: row ct*16 + ccol, segment kc*4 + g, swizzled
    int foff[KCH];
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) foff[kc] = ccol * 16 + ((kc * 4 + g) ^ ccol);

    for (int q0 = 0; q0 < Q; q0 += RT * 16) {
        // ---- this wave's caption: 6 row tiles x 4 K-chunks of A fragments, straight from global memory ----
        bf16x8 af[RT][KCH];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bf16x8* rowp = reinterpret_cast<const bf16x8*>(txt + ((size_t)bc * Q + min(q0 + rt * 16 + ccol, Q - 1)) * d + g * 8);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) af[rt][kc] = rowp[kc * 4];
        }
        // query-side keep bits of this lane's accumulator rows (row = 16 rt + 4 g + n) and, per row tile, whether any
        // lane of the wave sees a masked row (wave-uniform: the unmasked tiles skip the selects)
        unsigned tkeep = 0xffffffu, rt_masked = 0;
        if (tmask) {
            tkeep = 0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    tkeep |= (tmask[(size_t)bc * Q + min(q0 + rt * 16 + crow + n, Q - 1)] ? 1u : 0u) << (rt * 4 + n);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                if (__builtin_amdgcn_ballot_w64(((tkeep >> (rt * 4)) & 15u) != 15u) != 0) rt_masked |= 1u << rt;
        }

        const int qlim = Q - (q0 + crow + ccol);   // row 16 rt + crow + ccol of this pass exists iff 16 rt < qlim
        uint4 xs[NS] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        unsigned ck = 1;
        if (n_img > 0) {
            stage_load(a0, xs, ck);
            stage_write(0, xs, ck);
            if (n_img > 1) stage_load(a0 + 1, xs, ck);
        }
        __syncthreads();
        f32x4 rmx[RT];
        int rix[ARGS ? RT : 1][4];   // ARGS: column tile that holds the running row maximum (first one on ties: ct ascends)
        auto row_epilogue = [&](int a) {   // row maxima of image a: one 16-lane butterfly per accumulator register
            if (!(b < B && out_maxV)) return;
            const size_t at0 = ((size_t)bc * A + a) * Q + q0 + crow + ccol;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                float m0 = rmx[rt][0], m1 = rmx[rt][1], m2 = rmx[rt][2], m3 = rmx[rt][3];
                const float o0 = m0, o1 = m1, o2 = m2, o3 = m3;
                VLG_AM_DPP4("v_max_f32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
                VLG_AM_DPP4("v_max_f32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
                VLG_AM_DPP4("v_max_f32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf");
                VLG_AM_DPP4("v_max_f32_dpp", "row_mirror row_mask:0xf bank_mask:0xf");
                float res = m3;   // lane (g, ccol < 4) keeps row 16 rt + 4 g + ccol
                res = is2 ? m2 : res;
                res = is1 ? m1 : res;
                res = is0 ? m0 : res;
                if (ARGS) {
                    // where: the smallest region index among the lanes that hold the row maximum (a second, integer butterfly)
                    int m0i = o0 == m0 ? rix[ARGS ? rt : 0][0] * 16 + ccol : 0x7fff, m1i = o1 == m1 ? rix[ARGS ? rt : 0][1] * 16 + ccol : 0x7fff;
                    int m2i = o2 == m2 ? rix[ARGS ? rt : 0][2] * 16 + ccol : 0x7fff, m3i = o3 == m3 ? rix[ARGS ? rt : 0][3] * 16 + ccol : 0x7fff;
                    {
                        int &m0 = m0i, &m1 = m1i, &m2 = m2i, &m3 = m3i;   // the macro names its operands m0..m3
                        VLG_AM_DPP4("v_min_i32_dpp", "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
                        VLG_AM_DPP4("v_min_i32_dpp", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
                        VLG_AM_DPP4("v_min_i32_dpp", "row_half_mirror row_mask:0xf bank_mask:0xf");
                        VLG_AM_DPP4("v_min_i32_dpp", "row_mirror row_mask:0xf bank_mask:0xf");
                    }
                    int ri = m3i;
                    ri = is2 ? m2i : ri;
                    ri = is1 ? m1i : ri;
                    ri = is0 ? m0i : ri;
                    if (ccol < 4 && rt * 16 < qlim) xa.argV[at0 + rt * 16] = (uint16_t)ri;
                }
                if (ccol < 4 && rt * 16 < qlim) out_maxV[at0 + rt * 16] = res;
            }
        };
        for (int i = 0; i < n_img; ++i) {
            const int a = a0 + i, buf = i & 1;
            if (i + 1 < n_img) stage_write(buf ^ 1, xs, ck);       // tile i+1: loaded during the previous image's MFMAs
            if (i + 2 < n_img) stage_load(a + 2, xs, ck);          // tile i+2: lands during this image's MFMAs
            const uint4* tb = tiles[buf];
            float* dstQ = out_maxQ + ((size_t)bc * A + a) * V;
            unsigned ckl = 7u;   // region-side keep bits of this lane's three columns
            bool col_masked = false;
            if (vmask) {
                ckl = (unsigned)ckeep_s[buf][ccol] | ((unsigned)ckeep_s[buf][16 + ccol] << 1) | ((unsigned)ckeep_s[buf][32 + ccol] << 2);
                col_masked = __builtin_amdgcn_ballot_w64(ckl != 7u) != 0;
            }
            // column tile outermost: only its four B fragments are live; the element-wise running maximum over the column
            // tiles (-> row maxima) is kept per row tile, the one over the row tiles (-> column maxima) per column tile
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) rmx[rt] = f32x4{ninf, ninf, ninf, ninf};
            const bool prior_on = ARGS && xa.pen != nullptr && a == b;   // wave-uniform: the diagonal pair of this caption
#pragma unroll 1
            for (int ct = 0; ct < 3; ++ct) {
                bf16x8 bfr[KCH];
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) bfr[kc] = *reinterpret_cast<const bf16x8*>(tb + ct * 256 + foff[kc]);
                f32x4 cmx = f32x4{ninf, ninf, ninf, ninf};
                int cix[4] = {0, 0, 0, 0};   // ARGS: row tile that holds the running column maximum (first one: rt ascends)
                const unsigned ckc = (ckl >> ct) & 1u;
                const float* prow = nullptr;   // prior row of this lane's column segment
                if (prior_on) prow = xa.pen + (size_t)bc * Q * xa.n_seg + xa.seg_of_v[min(ct * 16 + ccol, V - 1)];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[rt][kc], bfr[kc], acc, 0, 0, 0);
                    if (col_masked || ((rt_masked >> rt) & 1u)) {   // wave-uniform: masked_fill_ of joint.py:417-418
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[n] = ((tkeep >> (rt * 4 + n)) & ckc) ? acc[n] : neg_inf;
                    }
                    if (prior_on) {   // joint.py:466-469
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[n] -= prow[(size_t)min(q0 + rt * 16 + crow + n, Q - 1) * xa.n_seg];
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if (ARGS) {   // strict >: the earlier column tile / row tile keeps a tie
                            const bool up = acc[n] > rmx[rt][n];
                            rmx[rt][n] = up ? acc[n] : rmx[rt][n];
                            rix[ARGS ? rt : 0][n] = up ? ct : rix[ARGS ? rt : 0][n];
                            const bool upc = acc[n] > cmx[n];
                            cmx[n] = upc ? acc[n] : cmx[n];
                            cix[n] = upc ? rt : cix[n];
                        } else {
                            rmx[rt][n] = fmaxf(rmx[rt][n], acc[n]);
                            cmx[n] = fmaxf(cmx[n], acc[n]);
                        }
                    }
                }
                if (out_maxQ && !ARGS) {   // column maxima of this column tile: within-lane over the four rows, then across the row groups
                    float m = fmaxf(fmaxf(cmx[0], cmx[1]), fmaxf(cmx[2], cmx[3]));
                    // lanes l, l^16, l^32, l^48 hold the same column: v_permlane16/32_swap with both operands = m return
                    // (own, partner) in some order on every lane -- no LDS round trip (ds_bpermute costs one per step)
                    {
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                        m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                    }
                    {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                        m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
                    }
                    const int v = ct * 16 + ccol;
                    if (b < B && lane < 16 && v < V) {
                        float* dst = dstQ + v;
                        *dst = q0 == 0 ? m : fmaxf(*dst, m);   // the same wave handles every row group of (b, a)
                    }
                }
                if (out_maxQ && ARGS) {
                    // (value, query) pairs: the larger value wins, equal values keep the smaller query.  Within the lane the four
                    // rows 16 cix + 4 g + n; then the other row groups g (same column) by v_permlane16/32_swap of both halves.
                    float m = cmx[0];
                    int qi = cix[0] * 16 + crow;
#pragma unroll
                    for (int n = 1; n < 4; ++n) {
                        const int qn = cix[n] * 16 + crow + n;
                        const bool take = cmx[n] > m || (cmx[n] == m && qn < qi);
                        m = take ? cmx[n] : m;
                        qi = take ? qn : qi;
                    }
                    qi += q0;
#pragma unroll
                    for (int step = 0; step < 2; ++step) {
                        const auto rv = step == 0 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false)
                                                  : __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
                        const auto ri = step == 0 ? __builtin_amdgcn_permlane16_swap((unsigned)qi, (unsigned)qi, false, false)
                                                  : __builtin_amdgcn_permlane32_swap((unsigned)qi, (unsigned)qi, false, false);
                        // (r[0], r[1]) = (own, partner) in some order, the same order for both swaps: fold both halves
                        const float va = __uint_as_float(rv[0]), vb = __uint_as_float(rv[1]);
                        const int ia = (int)ri[0], ib = (int)ri[1];
                        const bool tb_ = vb > va || (vb == va && ib < ia);
                        m = tb_ ? vb : va;
                        qi = tb_ ? ib : ia;
                    }
                    const int v = ct * 16 + ccol;
                    if (b < B && lane < 16 && v < V) {
                        const size_t at = ((size_t)bc * A + a) * V + v;
                        if (q0 == 0 || m > out_maxQ[at]) {   // later row groups only win with a strictly larger value
                            out_maxQ[at] = m;
                            xa.argQ[at] = (uint16_t)qi;
                        }
                    }
                }
            }
            row_epilogue(a);
#ifndef VLG_ABL_AM_NOBARRIER   // tools/ ablation: what the per-image barrier costs (results are wrong without it)
            __syncthreads();
#endif
        }
    }
}
#undef VLG_AM_DPP4

// =====================================================================================================
// The grounding loss's maxima WITH their positions (joint.py:446-483), round 3.  align_max_kernel<true, 3> above spends its
// time on the vector ALU, not on the matrix cores: per image and 48-row pass 36 MFMAs (576 cycles) against ~520 vector
// instructions (2100 cycles) -- (compare, select, select) per element and direction to carry a position next to each running
// maximum, and for the maxima over regions a 16-lane DPP butterfly per accumulator register, once for the value and once for
// the position -- with two wavefronts per SIMD to hide the dependent DPP chains behind (217 registers): 267 us at config-2.
// Here every score is computed TWICE, S = txt vis^T and S^T = vis txt^T -- the same two fragment sets with the MFMA's operands
// exchanged, so no extra loads -- because the accumulator layout (a lane holds four ROWS of one column) makes a maximum over
// the tile's rows a within-lane operation: S gives a lane queries of one region (max over Q), S^T gives it regions of one query
// (max over V).  Both directions are then: within-lane maximum (v_max3), two v_permlane swaps across the four row groups, and
// the position found AFTER the maximum is known -- first element equal to it, (compare, select) per element, smallest position
// across the row groups.  No 16-lane butterflies, all 96 query rows in one pass (the position registers are gone), and the
// matrix cores -- idle 80 % of the time before -- carry the second product.
// =====================================================================================================
// (fmed3(a, b, +inf) = max(a, b) without the canonicalising v_max x, x, x that fmaxf puts in front of MFMA results; chains fold to v_max3)
static __device__ __forceinline__ float am_max(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, __builtin_inff()); }
static __device__ __forceinline__ float am_max3(float a, float b, float c) { return am_max(am_max(a, b), c); }
// Maximum / minimum over the four row groups of a column (lanes l, l ^ 16, l ^ 32, l ^ 48), every lane gets it: v_permlane16/32_swap
// of a copy exchange the halves, no LDS round trip.  Hand-scheduled: left to itself hipcc canonicalises both swap results before
// each maximum (12 instructions against 8).  Operands are results of vector-ALU instructions (never of an MFMA directly: inline
// assembly is invisible to the compiler's MFMA hazard padding).
static __device__ __forceinline__ float am_xg_max(float m) {
    float t;
    asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1\n\t"
        "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_max_f32 %0, %0, %1"
        : "+v"(m), "=&v"(t));
    return m;
}
static __device__ __forceinline__ unsigned am_xg_min(unsigned m) {
    unsigned t;
    asm("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_min_u32 %0, %0, %1\n\t"
        "v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_min_u32 %0, %0, %1"
        : "+v"(m), "=&v"(t));
    return m;
}
// r <- code C_p of the FIRST p with x_p == m (r unchanged when none): compares into lane masks four at a time, then their selects,
// in descending p.  Written out because hipcc turns every formulation of this into (compare, wait two states for the mask,
// select) pairs in one serial chain.  x_p must have been read by a compiler-visible vector instruction before (they have: m is
// their maximum), so that an MFMA that produced them is known to have finished.
template <int C0, int C1, int C2, int C3, int C4, int C5, int C6, int C7, int C8, int C9, int C10, int C11>
static __device__ __forceinline__ unsigned am_first_eq12(unsigned r, float m, float x0, float x1, float x2, float x3, float x4, float x5,
                                                         float x6, float x7, float x8, float x9, float x10, float x11) {
    unsigned long long s0, s1, s2, s3;   // four masks in flight: a select reads the mask written four instructions earlier
    asm("v_cmp_eq_f32_e64 %[s3], %[x11], %[m]\n\tv_cmp_eq_f32_e64 %[s2], %[x10], %[m]\n\tv_cmp_eq_f32_e64 %[s1], %[x9], %[m]\n\tv_cmp_eq_f32_e64 %[s0], %[x8], %[m]\n\tv_cndmask_b32_e64 %[r], %[r], %[c11], %[s3]\n\tv_cndmask_b32_e64 %[r], %[r], %[c10], %[s2]\n\tv_cndmask_b32_e64 %[r], %[r], %[c9], %[s1]\n\tv_cndmask_b32_e64 %[r], %[r], %[c8], %[s0]\n\tv_cmp_eq_f32_e64 %[s3], %[x7], %[m]\n\tv_cmp_eq_f32_e64 %[s2], %[x6], %[m]\n\tv_cmp_eq_f32_e64 %[s1], %[x5], %[m]\n\tv_cmp_eq_f32_e64 %[s0], %[x4], %[m]\n\tv_cndmask_b32_e64 %[r], %[r], %[c7], %[s3]\n\tv_cndmask_b32_e64 %[r], %[r], %[c6], %[s2]\n\tv_cndmask_b32_e64 %[r], %[r], %[c5], %[s1]\n\tv_cndmask_b32_e64 %[r], %[r], %[c4], %[s0]\n\tv_cmp_eq_f32_e64 %[s3], %[x3], %[m]\n\tv_cmp_eq_f32_e64 %[s2], %[x2], %[m]\n\tv_cmp_eq_f32_e64 %[s1], %[x1], %[m]\n\tv_cmp_eq_f32_e64 %[s0], %[x0], %[m]\n\tv_cndmask_b32_e64 %[r], %[r], %[c3], %[s3]\n\tv_cndmask_b32_e64 %[r], %[r], %[c2], %[s2]\n\tv_cndmask_b32_e64 %[r], %[r], %[c1], %[s1]\n\tv_cndmask_b32_e64 %[r], %[r], %[c0], %[s0]"
        : [r] "+v"(r), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3)
        : [m] "v"(m), [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [x4] "v"(x4), [x5] "v"(x5), [x6] "v"(x6), [x7] "v"(x7), [x8] "v"(x8), [x9] "v"(x9), [x10] "v"(x10), [x11] "v"(x11),
          [c0] "n"(C0), [c1] "n"(C1), [c2] "n"(C2), [c3] "n"(C3), [c4] "n"(C4), [c5] "n"(C5), [c6] "n"(C6), [c7] "n"(C7), [c8] "n"(C8), [c9] "n"(C9), [c10] "n"(C10), [c11] "n"(C11));
    return r;
}

// x where bit `pos` of `bits` is set, `fill` elsewhere: a sign-extending one-bit extract is the select mask of v_bitop3 (a ? b : c =
// table 0xca) -- two vector instructions and no lane mask in SGPRs (a v_cmp / v_cndmask pair waits two states on the mask, and hipcc
// keeps 24 of them live).  The builtin, because hipcc rewrites (x & m) | (fill & ~m) with m known to be 0 / -1 as compare + select.
static __device__ __forceinline__ float am_keep(unsigned bits, int pos, float x, float fill) {
    const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)bits, pos, 1);
    return __uint_as_float(__builtin_amdgcn_bitop3_b32(mk, __float_as_uint(x), __float_as_uint(fill), 0xca));
}
static __device__ __forceinline__ unsigned am_keep_u(unsigned bits, int pos, unsigned x, unsigned fill) {
    const unsigned mk = (unsigned)__builtin_amdgcn_sbfe((int)bits, pos, 1);
    return __builtin_amdgcn_bitop3_b32(mk, x, fill, 0xca);
}
// (wave-uniform branches whose bodies are a few selects are flattened by hipcc into selects on EVERY tile; an empty volatile
//  statement inside keeps them branches)
#define VLG_AM_KEEP_BRANCH() asm volatile("" ::: "memory")
// After a group of MFMAs whose results are first read on the far side of a wave-uniform branch: hipcc pads MFMA -> vector-ALU reads
// with s_nop within a block, but with the branch in between the reader at the branch TARGET came out two instructions after the
// last MFMA (measured: wrong maxima in the unmasked variant only, right again with this pad; 16 states cover a 16-pass MFMA).
// The accumulators are tied operands: the pad can neither be scheduled ahead of the MFMAs that write them nor behind their first reader.
#define VLG_AM_MFMA_DRAIN3(A, B, C) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(A), "+v"(B), "+v"(C))
#define VLG_AM_MFMA_DRAIN6(A, B, C, D, E, F) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E), "+v"(F))

// ARGS = false: the maxima alone (the decoder's / gather_logit's fused dispatch) -- no searches, no position stores
// Several region groups per image (V > 48, the shipped factor layout): the staged tile is one GROUP of 48 regions, the loop runs over
// (image, group) steps; a region belongs to one group, so the maxima over the queries are per step as before, while the maxima over
// the regions are carried across an image's groups (strictly larger wins: the earlier group keeps a tie) and stored after its last one.
// NP = 2 (round 5): float32 features as two fp16 parts each under ONE power-of-two scale per tensor (align_split_kernel: hi = fp16(x s),
// lo = fp16(x s - hi); txt / vis = the hi parts, parts.txt_lo / vis_lo the lo parts) -- a product is hi hi + hi lo + lo hi, three
// v_mfma_f32_16x16x32_f16 into one accumulator, 2^-22 relative: float32's own level -- where align_mfma_kernel<fp32> runs on
// v_mfma_f32_16x16x4_f32 at 1/16 of the rate (1.18 ms at config-2).  48 queries per pass (the two parts of 96 would need 190 fragment
// registers); comparisons run on the scaled scores (a power of two: same order, same ties), the stored maxima are multiplied by
// 1 / (s_txt s_vis) (parts.inv, written by the split), a masked maximum stays neg_inf.
struct AlignParts {
    const uint16_t *txt_lo, *vis_lo;
    const float* inv;   // device: {1 / s_txt, 1 / s_vis}
};
template <int NP>
static __device__ __forceinline__ f32x4 am_mma(const bf16x8& a, const bf16x8& b, f32x4 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
    if constexpr (NP == 2) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool HASQ, bool ARGS, bool MULTI, int NP = 1>   // MULTI = false: one group (V <= 48), the group arithmetic folds away
__global__ __launch_bounds__(kAMThreads) void align_argmax_kernel(
    const uint16_t* __restrict__ txt, const uint16_t* __restrict__ vis, const uint8_t* __restrict__ tmask,
    const uint8_t* __restrict__ vmask, int B, int A, int Q, int V, float neg_inf, float* __restrict__ out_maxV,
    float* __restrict__ out_maxQ, int a_per_block, AlignArgs xa, AlignParts parts) {
    constexpr int d = 128, KCH = 4, RT = NP == 2 ? 3 : 6;
    constexpr unsigned BIG = 0x7000u;
    __shared__ uint4 tiles[2][NP][kAMSlots];
    float unscale = 1.f;
    if constexpr (NP == 2) unscale = parts.inv[0] * parts.inv[1];
    auto fin = [&](float m) { return NP == 2 ? (m == neg_inf ? neg_inf : m * unscale) : m; };   // a stored maximum
    __shared__ unsigned long long kmask_s[2];   // region-side keep bits of the staged image (bit v), by wave 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: everything that depends on the caption lives in SGPRs
    const int b = blockIdx.y * kAMWaves + wave, bc = min(b, B - 1);   // a wave past the batch mirrors the last caption, stores nothing
    const int a0 = blockIdx.x * a_per_block, n_img = min(A, a0 + a_per_block) - a0;
    const int NG = MULTI ? (V + kAMRows - 1) / kAMRows : 1, n_step = n_img * NG;   // region groups per image; (image, group) steps of this block
    const int g = lane >> 4, ccol = lane & 15, crow = g * 4;
    constexpr int NS = 2;
    const bool has2 = tid < kAMSlots - kAMThreads;
    int srow[NS];
    unsigned sseg[NS];   // this thread's slots of a tile: row within the group, byte offset of the (swizzled) segment within a row
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int sl = tid + k * kAMThreads, r = sl >> 4;
        srow[k] = r;
        sseg[k] = 2u * (unsigned)(((sl & 15) ^ (r & 15)) * 8);
    }
    auto stage_load = [&](int st, uint4* x, unsigned& ck) {   // step st = (image, group); x[part * NS + slot]
        const int a = MULTI ? a0 + st / NG : a0 + st, v0 = MULTI ? (st % NG) * kAMRows : 0;
        // (32-bit offsets from a scalar base; rows past V are copies of the last region: they cannot change a maximum, lose every tie)
        const unsigned o0 = 2u * (unsigned)(min(v0 + srow[0], V - 1) * d) + sseg[0], o1 = 2u * (unsigned)(min(v0 + srow[1], V - 1) * d) + sseg[1];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            const char* img = reinterpret_cast<const char*>((pp == 0 ? vis : parts.vis_lo) + (size_t)a * V * d);
            x[pp * NS] = *reinterpret_cast<const uint4*>(img + o0);
            if (has2) x[pp * NS + 1] = *reinterpret_cast<const uint4*>(img + o1);
        }
        if (vmask && tid < kAMRows) ck = vmask[(size_t)a * V + min(v0 + tid, V - 1)];
    };
    auto stage_write = [&](int buf, const uint4* x, unsigned ck) {
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            tiles[buf][pp][tid] = x[pp * NS];
            if (has2) tiles[buf][pp][tid + kAMThreads] = x[pp * NS + 1];
        }
        if (vmask && wave == 0) {   // (lanes 48.. hold ck = 1)
            const unsigned long long km = __builtin_amdgcn_ballot_w64(ck != 0);
            if (lane == 0) kmask_s[buf] = km;
        }
    };
    int foff[KCH];
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) foff[kc] = ccol * 16 + ((kc * 4 + g) ^ ccol);

    for (int q0 = 0; q0 < Q; q0 += RT * 16) {   // one pass up to 96 queries
        bf16x8 af[NP][RT][KCH];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const bf16x8* rowp = reinterpret_cast<const bf16x8*>((pp == 0 ? txt : parts.txt_lo) + ((size_t)bc * Q + min(q0 + rt * 16 + ccol, Q - 1)) * d + g * 8);
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) af[pp][rt][kc] = rowp[kc * 4];
            }
        // Masks (masked_fill_ of joint.py:417-418) are applied per element only along the axis a maximum runs over -- queries in
        // S (rows 16 rt + 4 g + n: bit 4 rt + n of tkeep; only the row tiles that have a masked query, rt_masked, wave-uniform),
        // regions in S^T -- and to the RESULT for the other axis: a masked query's row of S^T is all neg_inf, so its maximum
        // is neg_inf at position 0 (tkeepT: bit rt = query 16 rt + ccol), likewise a masked region's column of S.
        unsigned tkeep = (1u << (4 * RT)) - 1u, tkeepT = (1u << RT) - 1u, rt_masked = 0;
        if (tmask) {
            tkeep = 0;
            tkeepT = 0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    tkeep |= (tmask[(size_t)bc * Q + min(q0 + rt * 16 + crow + n, Q - 1)] ? 1u : 0u) << (rt * 4 + n);
                tkeepT |= (tmask[(size_t)bc * Q + min(q0 + rt * 16 + ccol, Q - 1)] ? 1u : 0u) << rt;
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
                if (__builtin_amdgcn_ballot_w64(((tkeep >> (rt * 4)) & 15u) != 15u) != 0) rt_masked |= 1u << rt;
        }
        const bool t_any = rt_masked != 0;   // (wave-uniform)
        uint4 xs[NP * NS];
#pragma unroll
        for (int k = 0; k < NP * NS; ++k) xs[k] = make_uint4(0, 0, 0, 0);
        unsigned ck = 1;
        if (n_step > 0) {
            stage_load(0, xs, ck);
            stage_write(0, xs, ck);
            if (n_step > 1) stage_load(1, xs, ck);
        }
        __syncthreads();
        float mrow[RT];       // maxima over the regions (and where), carried across an image's groups
        unsigned irow[RT];
        for (int i = 0; i < n_step; ++i) {
            const int a = MULTI ? a0 + i / NG : a0 + i, grp = MULTI ? i % NG : 0, v0 = grp * kAMRows, buf = i & 1;
            if (i + 1 < n_step) stage_write(buf ^ 1, xs, ck);       // tile i+1: loaded during the previous step's MFMAs
            if (i + 2 < n_step) stage_load(i + 2, xs, ck);          // tile i+2: lands during this step's MFMAs
            const uint4* tb = tiles[buf][0];
            // region-side keep bits: S column 16 ct + ccol (bit ct of ckl), S^T rows 16 ct + 4 g + n (bit 4 ct + n of cklT);
            // the image's 48-bit mask is wave-uniform, the per-lane views are only built for an image that has masked regions
            unsigned ckl = 7u, cklT = 0xfffu;
            bool v_any = false;
            if (vmask) {
                const unsigned long long kmv = kmask_s[buf];
                const unsigned klo = __builtin_amdgcn_readfirstlane((unsigned)kmv), khi = __builtin_amdgcn_readfirstlane((unsigned)(kmv >> 32));
                v_any = klo != 0xffffffffu || (khi & 0xffffu) != 0xffffu;
                if (v_any) {
                    VLG_AM_KEEP_BRANCH();
                    ckl = ((klo >> ccol) & 1u) | (((klo >> (16 + ccol)) & 1u) << 1) | (((khi >> ccol) & 1u) << 2);
                    cklT = ((klo >> crow) & 15u) | (((klo >> (16 + crow)) & 15u) << 4) | (((khi >> crow) & 15u) << 8);
                }
            }
            float* const rowV = out_maxV + ((size_t)bc * A + a) * Q + q0;          // wave-uniform bases
            uint16_t* const rowA = ARGS ? xa.argV + ((size_t)bc * A + a) * Q + q0 : nullptr;
            // NP = 1: the image's fragments stay in registers for both products (48).  NP = 2: hi | lo of all three region tiles would be 96
            // registers next to the 96 of the caption: a region tile's fragments are read when it is used (once per product), and the S^T
            // tiles of the pass (36 accumulators) are computed region tile by region tile before their maxima are taken query tile by query tile.
            bf16x8 bfr[NP == 1 ? 3 : 1][KCH];
            if constexpr (NP == 1) {
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) bfr[ct][kc] = *reinterpret_cast<const bf16x8*>(tb + ct * 256 + foff[kc]);
            }
            auto frag = [&](int pp, int ct, int kc) { return *reinterpret_cast<const bf16x8*>(tb + pp * kAMSlots + ct * 256 + foff[kc]); };
            f32x4 stA[NP == 2 ? RT : 1][3];
            if constexpr (NP == 2) {
                if (out_maxV) {
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) {
                        bf16x8 bh[KCH], bl[KCH];
#pragma unroll
                        for (int kc = 0; kc < KCH; ++kc) {
                            bh[kc] = frag(0, ct, kc);
                            bl[kc] = frag(1, ct, kc);
                        }
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) {
                            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int kc = 0; kc < KCH; ++kc) {   // the two small terms first
                                t = am_mma<NP>(bl[kc], af[0][rt][kc], t);
                                t = am_mma<NP>(bh[kc], af[NP - 1][rt][kc], t);
                                t = am_mma<NP>(bh[kc], af[0][rt][kc], t);
                            }
                            stA[rt][ct] = t;
                        }
                    }
                }
            }
            // ---- maxima over the regions: S^T, one query tile at a time ----
            // (tools/time_argmax_ablation.sh: -DVLG_ABL_AM_NOST / _NOS drop one of the two products, _NOSEARCH the first-equal searches -- wrong
            //  results, measured ceilings: HISTORY.md section 3.1a)
#ifdef VLG_ABL_AM_NOST
            if (false) {
#else
            if (out_maxV) {   // (kernel-uniform)
#endif
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                f32x4 st[3];   // rows = regions 16 ct + 4 g + n, column = query 16 rt + ccol
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    if constexpr (NP == 2) {
                        st[ct] = stA[rt][ct];
                    } else {
                        f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kc = 0; kc < KCH; ++kc) t = am_mma<NP>(bfr[ct][kc], af[0][rt][kc], t);
                        st[ct] = t;
                    }
                }
                VLG_AM_MFMA_DRAIN3(st[0], st[1], st[2]);
                if (v_any) {   // wave-uniform: this image has masked regions
                    VLG_AM_KEEP_BRANCH();
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                        for (int n = 0; n < 4; ++n) st[ct][n] = am_keep(cklT, ct * 4 + n, st[ct][n], neg_inf);
                }
                // twelve values in this lane, then the other three row groups; where: the first of this lane's regions that holds
                // the maximum, then the smallest position across the row groups
                const float t3 = am_max3(st[2][1], st[2][2], st[2][3]);   // (a repeated operand instead of a two-input maximum: that one gets canonicalised inputs)
                float m = am_max3(am_max3(st[0][0], st[0][1], st[0][2]), am_max3(st[0][3], st[1][0], st[1][1]),
                                  am_max3(am_max3(st[1][2], st[1][3], st[2][0]), t3, t3));
                m = am_xg_max(m);
                unsigned vi = 0;
#ifdef VLG_ABL_AM_NOSEARCH
                if (false) {
#else
                if (ARGS) {
#endif
                    vi = am_first_eq12<0, 1, 2, 3, 16, 17, 18, 19, 32, 33, 34, 35>(BIG, m, st[0][0], st[0][1], st[0][2], st[0][3], st[1][0], st[1][1],
                                                                                    st[1][2], st[1][3], st[2][0], st[2][1], st[2][2], st[2][3]);
                    vi = am_xg_min(vi + (unsigned)crow);
                }
                if (grp == 0) {   // (wave-uniform)
                    mrow[rt] = m;
                    irow[rt] = vi;
                } else {
                    VLG_AM_KEEP_BRANCH();
                    const bool up = m > mrow[rt];   // strictly: the earlier group keeps a tie
                    mrow[rt] = up ? m : mrow[rt];
                    if (ARGS) irow[rt] = up ? vi + (unsigned)v0 : irow[rt];
                }
                if (t_any && grp == NG - 1) {   // wave-uniform: a masked query's maximum
                    VLG_AM_KEEP_BRANCH();
                    mrow[rt] = am_keep(tkeepT, rt, mrow[rt], neg_inf);
                    if (ARGS) irow[rt] = am_keep_u(tkeepT, rt, irow[rt], 0u);
                }
            }
            // every lane of a column holds the six results of its column: row group g stores query tiles g and 4 + g, so that
            // a store instruction writes 64 (32) consecutive queries
            if (b < B && grp == NG - 1) {
                float m_lo;
                unsigned i_lo;
                if constexpr (RT == 6) {
                    m_lo = g == 0 ? mrow[0] : g == 1 ? mrow[1] : g == 2 ? mrow[2] : mrow[RT - 3];
                    i_lo = g == 0 ? irow[0] : g == 1 ? irow[1] : g == 2 ? irow[2] : irow[RT - 3];
                } else {   // (three tiles: row group 3 stores nothing)
                    const float m01 = g == 0 ? mrow[0] : mrow[1];
                    const unsigned i01 = g == 0 ? irow[0] : irow[1];
                    m_lo = g >= 2 ? mrow[2] : m01;
                    i_lo = g >= 2 ? irow[2] : i01;
                }
                if (lane < 16 * RT && q0 + lane < Q) {
                    rowV[lane] = fin(m_lo);
                    if (ARGS) rowA[lane] = (uint16_t)i_lo;
                }
                if constexpr (RT == 6) {
                    const float m_hi = g == 0 ? mrow[RT - 2] : mrow[RT - 1];
                    const unsigned i_hi = g == 0 ? irow[RT - 2] : irow[RT - 1];
                    if (lane < 32 && q0 + 64 + lane < Q) {
                        rowV[64 + lane] = fin(m_hi);
                        if (ARGS) rowA[64 + lane] = (uint16_t)i_hi;
                    }
                }
            }
            }   // out_maxV
            // ---- maxima over the queries: S, one region tile at a time ----
#ifdef VLG_ABL_AM_NOS
            if (false) {
#else
            if (HASQ) {
#endif
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) {
                    f32x4 sq[RT];   // rows = queries 16 rt + 4 g + n, column = region 16 ct + ccol
                    bf16x8 bh[KCH], bl[NP == 2 ? KCH : 1];
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) {
                        if constexpr (NP == 2) {
                            bh[kc] = frag(0, ct, kc);
                            bl[kc] = frag(1, ct, kc);
                        } else {
                            bh[kc] = bfr[ct][kc];
                        }
                    }
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int kc = 0; kc < KCH; ++kc) {
                            if constexpr (NP == 2) {
                                acc = am_mma<NP>(af[NP - 1][rt][kc], bh[kc], acc);
                                acc = am_mma<NP>(af[0][rt][kc], bl[kc], acc);
                            }
                            acc = am_mma<NP>(af[0][rt][kc], bh[kc], acc);
                        }
                        sq[rt] = acc;
                    }
                    if constexpr (RT == 6) VLG_AM_MFMA_DRAIN6(sq[0], sq[1], sq[2], sq[3], sq[4], sq[5]);
                    else VLG_AM_MFMA_DRAIN3(sq[0], sq[1], sq[2]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        if ((rt_masked >> rt) & 1u) {   // wave-uniform: this row tile has a masked query
                            VLG_AM_KEEP_BRANCH();
#pragma unroll
                            for (int n = 0; n < 4; ++n) sq[rt][n] = am_keep(tkeep, rt * 4 + n, sq[rt][n], neg_inf);
                        }
                    const float u0 = am_max3(am_max3(sq[0][0], sq[0][1], sq[0][2]), am_max3(sq[0][3], sq[1][0], sq[1][1]), am_max3(sq[1][2], sq[1][3], sq[2][0]));
                    float m;
                    if constexpr (RT == 6) {
                        const float u1 = am_max3(am_max3(sq[2][1], sq[2][2], sq[2][3]), am_max3(sq[RT - 3][0], sq[RT - 3][1], sq[RT - 3][2]), am_max3(sq[RT - 3][3], sq[RT - 2][0], sq[RT - 2][1]));
                        const float u2 = am_max3(sq[RT - 2][2], sq[RT - 2][3], sq[RT - 1][0]), u3 = am_max3(sq[RT - 1][1], sq[RT - 1][2], sq[RT - 1][3]);
                        m = am_max3(u0, u1, am_max3(u2, u3, u3));
                    } else {
                        const float u1 = am_max3(sq[2][1], sq[2][2], sq[2][3]);
                        m = am_max3(u0, u1, u1);
                    }
                    m = am_xg_max(m);
                    // position code 4 rt + n (query 16 rt + 4 g + n; 80.. is no inline constant), later row tiles first
                    unsigned qi = 0;
#ifdef VLG_ABL_AM_NOSEARCH
                    if (false) {
#else
                    if (ARGS) {
#endif
                        unsigned qc = BIG;
                        if constexpr (RT == 6)
                            qc = am_first_eq12<12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23>(BIG, m, sq[RT - 3][0], sq[RT - 3][1], sq[RT - 3][2], sq[RT - 3][3], sq[RT - 2][0],
                                                                                                sq[RT - 2][1], sq[RT - 2][2], sq[RT - 2][3], sq[RT - 1][0], sq[RT - 1][1], sq[RT - 1][2], sq[RT - 1][3]);
                        qc = am_first_eq12<0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11>(qc, m, sq[0][0], sq[0][1], sq[0][2], sq[0][3], sq[1][0], sq[1][1], sq[1][2],
                                                                                 sq[1][3], sq[2][0], sq[2][1], sq[2][2], sq[2][3]);
                        qi = ((qc & ~3u) << 2) + (qc & 3u) + (unsigned)crow;   // BIG stays far above every query
                        qi = am_xg_min(qi) + (unsigned)q0;
                    }
                    if (v_any) {   // a masked region's maximum
                        VLG_AM_KEEP_BRANCH();
                        m = am_keep(ckl, ct, m, neg_inf);
                        if (ARGS) qi = am_keep_u(ckl, ct, qi, (unsigned)q0);
                    }
                    const int v = v0 + ct * 16 + ccol;
                    if (b < B && lane < 16 && v < V) {
                        float* const colV = out_maxQ + ((size_t)bc * A + a) * V;   // wave-uniform bases
                        const float mu = fin(m);
                        if (q0 == 0 || mu > colV[v]) {   // later passes (Q > 16 RT) only win with a strictly larger value
                            colV[v] = mu;
                            if (ARGS) xa.argQ[((size_t)bc * A + a) * V + v] = (uint16_t)qi;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
}

// The diagonal pairs (a == b) again, with the POS prior pen[b,q,seg(v)] subtracted before the maxima (joint.py:466-469): B
// pairs of 65 536 -- the scores go to LDS and one lane scans a row / a column in order (first position on ties, by construction).
// Runs after align_argmax_kernel on the same stream and overwrites its outputs for these pairs.
// NW wavefronts per pair (round 5): wavefront w takes the region groups w, w + NW, ... of the image, each into its own LDS score
// tile; a region belongs to one group, so the maxima over the queries are final per group, while every wavefront carries its own
// running maxima over the regions and the NW of them meet at the end (larger value, then SMALLER region: the first position).  With one
// wavefront the shipped factor layout's 29 groups of 48 columns were walked serially by 64 wavefronts on the whole chip: 404 us.
constexpr int kPdP = 49, kPdRT = 6;
template <int NP>   // NP = 2: float32 features on two fp16 parts (see align_argmax_kernel); the scores are unscaled BEFORE the prior is subtracted
__global__ __launch_bounds__(512) void align_prior_diag_kernel(
    const uint16_t* __restrict__ txt, const uint16_t* __restrict__ vis, const uint8_t* __restrict__ tmask,
    const uint8_t* __restrict__ vmask, int B, int A, int Q, int V, float neg_inf, float* __restrict__ out_maxV,
    float* __restrict__ out_maxQ, AlignArgs xa, AlignParts parts) {
    constexpr int d = 128, KCH = 4, RT = kPdRT, P = kPdP;
    extern __shared__ __attribute__((aligned(16))) float pd_smem[];   // [NW][RT 16 P] score tiles, then [NW][128] (max, position) pairs
    __shared__ uint8_t kq_s[RT * 16], kv_s[8][48];
    __shared__ float pen_s[RT * 16][8];   // this pass's rows of the prior table (n_seg <= 8): looked up from LDS beside the operand reads --
                                          // read from memory behind the MFMAs they were a second round trip per tile batch (66 % of the
                                          // kernel's wave cycles were s_waitcnt)
    const int NW = blockDim.x >> 6, wave = threadIdx.x >> 6;
    float* S = pd_smem + (size_t)wave * RT * 16 * P;
    float* mrg_m = pd_smem + (size_t)NW * RT * 16 * P;                 // [NW][128]
    int* mrg_i = reinterpret_cast<int*>(mrg_m + NW * 128);
    const int b = blockIdx.x, a = b, lane = threadIdx.x & 63, g = lane >> 4, ccol = lane & 15, crow = g * 4;
    if (a >= A) return;
    const float ninf = neg_infinity();
    const float* pen_b = xa.pen + (size_t)b * Q * xa.n_seg;
    const int NG = (V + 47) / 48, n_it = (NG + NW - 1) / NW;           // every wavefront runs n_it rounds (the barriers are workgroup-wide)
    for (int q0 = 0; q0 < Q; q0 += RT * 16) {
        __syncthreads();
        for (int i = threadIdx.x; i < RT * 16; i += blockDim.x) kq_s[i] = tmask ? (uint8_t)(tmask[(size_t)b * Q + min(q0 + i, Q - 1)] != 0) : (uint8_t)1;
        const bool pen_lds = xa.n_seg <= 8;   // (uniform)
        if (pen_lds)
            for (int i = threadIdx.x; i < RT * 16 * xa.n_seg; i += blockDim.x) {
                const int row = i / xa.n_seg, sg = i - row * xa.n_seg;
                pen_s[row][sg] = pen_b[(size_t)min(q0 + row, Q - 1) * xa.n_seg + sg];
            }
        const int nq = min(RT * 16, Q - q0);
        float m_run[2] = {ninf, ninf};   // maxima over the regions of queries lane, lane + 64: carried across this wavefront's region groups
        int vi_run[2] = {0, 0};
        for (int it = 0; it < n_it; ++it) {
            const int grp = it * NW + wave, v0 = grp * 48;
            const bool on = grp < NG;                                   // (wave-uniform)
            if (on && lane < 48) kv_s[wave][lane] = vmask ? (uint8_t)(vmask[(size_t)a * V + min(v0 + lane, V - 1)] != 0) : (uint8_t)1;
            __syncthreads();
            if (on) {
                int seg3[3];   // the factor segment of this lane's column in each of the three column tiles: once per round, not per row tile
#pragma unroll
                for (int ct = 0; ct < 3; ++ct) seg3[ct] = xa.seg_of_v[min(v0 + ct * 16 + ccol, V - 1)];
#pragma unroll 2
                for (int rt = 0; rt < RT; ++rt)   // (two row tiles x three column tiles of operand reads in flight: the kernel is a chain of round trips)
#pragma unroll
                    for (int ct = 0; ct < 3; ++ct) {
                        const bf16x8* ap = reinterpret_cast<const bf16x8*>(txt + ((size_t)b * Q + min(q0 + rt * 16 + ccol, Q - 1)) * d + g * 8);
                        const int v = min(v0 + ct * 16 + ccol, V - 1);
                        const bf16x8* bp = reinterpret_cast<const bf16x8*>(vis + ((size_t)a * V + v) * d + g * 8);
                        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
                        if constexpr (NP == 2) {
                            const bf16x8* al = reinterpret_cast<const bf16x8*>(parts.txt_lo + ((size_t)b * Q + min(q0 + rt * 16 + ccol, Q - 1)) * d + g * 8);
                            const bf16x8* bl = reinterpret_cast<const bf16x8*>(parts.vis_lo + ((size_t)a * V + v) * d + g * 8);
#pragma unroll
                            for (int kc = 0; kc < KCH; ++kc) {
                                acc = am_mma<2>(al[kc * 4], bp[kc * 4], acc);
                                acc = am_mma<2>(ap[kc * 4], bl[kc * 4], acc);
                            }
                        }
#pragma unroll
                        for (int kc = 0; kc < KCH; ++kc) acc = am_mma<NP>(ap[kc * 4], bp[kc * 4], acc);
                        if constexpr (NP == 2) {
                            const float un = parts.inv[0] * parts.inv[1];
#pragma unroll
                            for (int n = 0; n < 4; ++n) acc[n] *= un;
                        }
                        const unsigned vk = kv_s[wave][ct * 16 + ccol];
                        const float* pen_v = pen_b + seg3[ct];
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            const int q = min(q0 + rt * 16 + crow + n, Q - 1);
                            const bool keep = (vk & kq_s[rt * 16 + crow + n]) != 0;
                            const float pn = pen_lds ? pen_s[rt * 16 + crow + n][seg3[ct]] : pen_v[(size_t)q * xa.n_seg];
                            S[(rt * 16 + crow + n) * P + ct * 16 + ccol] = (keep ? acc[n] : neg_inf) - pn;
                        }
                    }
            }
            __syncthreads();
            if (on) {
                const int nv = min(48, V - v0);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = lane + 64 * h;
                    if (q < nq)
                        for (int v = 0; v < nv; ++v) {
                            const float x = S[q * P + v];
                            if (x > m_run[h]) { m_run[h] = x; vi_run[h] = v0 + v; }   // ascending groups within a wavefront: strict keeps the first
                        }
                }
                if (out_maxQ && lane < nv) {
                    const size_t at = ((size_t)b * A + a) * V + v0 + lane;
                    float m = q0 == 0 ? ninf : out_maxQ[at];
                    int qi = q0 == 0 ? 0 : (int)xa.argQ[at];
                    for (int q = 0; q < nq; ++q) {
                        const float x = S[q * P + lane];
                        if (x > m) { m = x; qi = q0 + q; }
                    }
                    out_maxQ[at] = m;
                    xa.argQ[at] = (uint16_t)qi;
                }
            }
        }
        if (out_maxV) {   // the NW running maxima of a query meet: larger value, then the smaller region (= the first position in order)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                mrg_m[wave * 128 + lane + 64 * h] = m_run[h];
                mrg_i[wave * 128 + lane + 64 * h] = vi_run[h];
            }
            __syncthreads();
            if (wave == 0)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int q = lane + 64 * h;
                    if (q < nq) {
                        float m = m_run[h];
                        int vi = vi_run[h];
                        for (int w = 1; w < NW; ++w) {
                            const float x = mrg_m[w * 128 + q];
                            const int xi = mrg_i[w * 128 + q];
                            if (x > m || (x == m && xi < vi)) { m = x; vi = xi; }
                        }
                        out_maxV[((size_t)b * A + a) * Q + q0 + q] = m;
                        xa.argV[((size_t)b * A + a) * Q + q0 + q] = (uint16_t)vi;
                    }
                }
        }
    }
}

// =====================================================================================================
// The materialised tensor attmap [B,A,Q,V] (what `gather_logit_simple` returns, joint.py:406-419) for one-group images
// (V <= 48, V % 4 == 0), bf16 features, d = 128 -- round 3.  align_mfma_kernel<TILE> keeps 96 caption rows AND a three-deep
// image-fragment ring in 286 registers (one wave per SIMD, four waves per CU), bounces every accumulator through an LDS
// tile (26 % of its LDS cycles are bank conflicts) and pays an LDS round trip per copy-out step: 0.228 ms = 3.4 TB/s of
// writes where a plain fill reaches 6.8.  This kernel is align_max_kernel's skeleton (image tiles shared by eight captions
// through the double-buffered swizzled LDS image, two waves per SIMD) with the MFMA operands SWAPPED: the tile computed is
// S^T[region][query], whose accumulator layout gives a lane four CONSECUTIVE regions of one query -- 16 contiguous bytes of
// the output row -- so every accumulator tile goes to the wave's LDS copy of the (b, a) output block as ONE ds_write_b128
// (18 per image instead of 72 ds_write_b32; pitch V floats = 144 bytes puts a 16-lane group on 64 distinct banks), and because
// that copy has the output's own layout the copy-out is linear: 12 conflict-free ds_read_b128 issued together, then 12 fully
// contiguous 1 KB stores.  (Storing the accumulators straight to global memory -- 16 rows x 64 bytes per instruction -- was
// measured first: 0.271 ms, write-combining of half cache lines is slower than the LDS detour.)
// =====================================================================================================
#ifndef VLG_AF_RT
#define VLG_AF_RT 6
#endif
#ifndef VLG_AF_WPE
#define VLG_AF_WPE 2
#endif
__global__ __launch_bounds__(kAMThreads, VLG_AF_WPE) void align_full_kernel(
    const uint16_t* __restrict__ txt, const uint16_t* __restrict__ vis, const uint8_t* __restrict__ tmask,
    const uint8_t* __restrict__ vmask, int B, int A, int Q, int V, float neg_inf, float* __restrict__ out_full, int a_per_block) {
    constexpr int d = 128, KCH = 4, RT = VLG_AF_RT;
    __shared__ uint4 tiles[2][kAMSlots];
    __shared__ uint8_t ckeep_s[2][kAMRows];
    extern __shared__ __attribute__((aligned(16))) float af_otile[];   // [8 waves][96 queries][V] fp32: the output block's own layout
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* otile = af_otile + (size_t)wave * RT * 16 * V;
    const int b = blockIdx.y * kAMWaves + wave, bc = min(b, B - 1);
    const int a0 = blockIdx.x * a_per_block, n_img = min(A, a0 + a_per_block) - a0;
    const int g = lane >> 4, ccol = lane & 15;
    constexpr int NS = 2;
    const bool has2 = tid < kAMSlots - kAMThreads;
    int soff[NS], srow[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int sl = tid + k * kAMThreads, r = sl >> 4;
        srow[k] = r;
        soff[k] = ((sl & 15) ^ (r & 15)) * 8;
    }
    auto stage_load = [&](int a, uint4* x, unsigned& ck) {
        const uint16_t* img = vis + (size_t)a * V * d;
        x[0] = *reinterpret_cast<const uint4*>(img + (size_t)min(srow[0], V - 1) * d + soff[0]);
        if (has2) x[1] = *reinterpret_cast<const uint4*>(img + (size_t)min(srow[1], V - 1) * d + soff[1]);
        if (vmask && tid < kAMRows) ck = vmask[(size_t)a * V + min(tid, V - 1)];
    };
    auto stage_write = [&](int buf, const uint4* x, unsigned ck) {
        tiles[buf][tid] = x[0];
        if (has2) tiles[buf][tid + kAMThreads] = x[1];
        if (vmask && tid < kAMRows) ckeep_s[buf][tid] = (uint8_t)(ck != 0);
    };
    int foff[KCH];
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) foff[kc] = ccol * 16 + ((kc * 4 + g) ^ ccol);

    for (int q0 = 0; q0 < Q; q0 += RT * 16) {
        const int n4 = (min(RT * 16, Q - q0) * V) >> 2;   // float4 of this pass's output block
        // this wave's caption: 6 query tiles x 4 K-chunks, the B operand of every MFMA (column = query q0 + 16 rt + ccol)
        bf16x8 qf[RT][KCH];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bf16x8* rowp = reinterpret_cast<const bf16x8*>(txt + ((size_t)bc * Q + min(q0 + rt * 16 + ccol, Q - 1)) * d + g * 8);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) qf[rt][kc] = rowp[kc * 4];
        }
        unsigned qkeep = (1u << RT) - 1u;   // bit rt: this lane's query of row tile rt is kept
        if (tmask) {
            qkeep = 0;
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) qkeep |= (tmask[(size_t)bc * Q + min(q0 + rt * 16 + ccol, Q - 1)] ? 1u : 0u) << rt;
        }
        const bool q_masked = __builtin_amdgcn_ballot_w64(qkeep != (1u << RT) - 1u) != 0;
        uint4 xs[NS] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        unsigned ck = 1;
        if (n_img > 0) {
            stage_load(a0, xs, ck);
            stage_write(0, xs, ck);
            if (n_img > 1) stage_load(a0 + 1, xs, ck);
        }
        __syncthreads();
        for (int i = 0; i < n_img; ++i) {
            const int a = a0 + i, buf = i & 1;
            if (i + 1 < n_img) stage_write(buf ^ 1, xs, ck);
            if (i + 2 < n_img) stage_load(a + 2, xs, ck);
            const uint4* tb = tiles[buf];
            float* ot = otile + ccol * V + 4 * g;                                        // + (16 rt) * V + 16 ct
            unsigned vkeep = 0xfffu;   // bit 4 ct + n: region 16 ct + 4 g + n is kept
            bool v_masked = false;
            if (vmask) {
                vkeep = 0;
#pragma unroll
                for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                    for (int n = 0; n < 4; ++n) vkeep |= (unsigned)ckeep_s[buf][ct * 16 + 4 * g + n] << (4 * ct + n);
                v_masked = __builtin_amdgcn_ballot_w64(vkeep != 0xfffu) != 0;
            }
            // (tools/time_align_full_ablation.sh: -DVLG_ABL_AF_NOMFMA drops the products and the LDS writes, -DVLG_ABL_AF_NOSTORE the global stores --
            //  wrong results, measured ceilings: HISTORY.md section 3)
#ifdef VLG_ABL_AF_NOMFMA
            if (false)
#endif
#pragma unroll 1
            for (int ct = 0; ct < 3; ++ct) {
                bf16x8 vf[KCH];   // A operand: regions 16 ct + (lane & 15) of the image tile
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) vf[kc] = *reinterpret_cast<const bf16x8*>(tb + ct * 256 + foff[kc]);
                const bool col_live = ct * 16 + 4 * g < V;   // V % 4 == 0: a lane's four regions are all inside or all outside
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kc], qf[rt][kc], acc, 0, 0, 0);
                    if (q_masked || v_masked) {   // wave-uniform: masked_fill_ of joint.py:417-418
                        const bool qk = (qkeep >> rt) & 1u;
#pragma unroll
                        for (int n = 0; n < 4; ++n) acc[n] = (qk && ((vkeep >> (4 * ct + n)) & 1u)) ? acc[n] : neg_inf;
                    }
                    if (col_live) *reinterpret_cast<float4*>(ot + (rt * 16) * V + ct * 16) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                }
            }
            // linear copy-out of the wave's block (same layout as the output): all reads first, then contiguous 1 KB stores
            __builtin_amdgcn_wave_barrier();
            if (b < B) {
#ifdef VLG_ABL_AF_SEQ   // (ablation: the same bytes, but the eight waves of a workgroup write NEIGHBOURING chunks instead of chunks 3 MB apart)
#ifndef VLG_ABL_AF_SEQ_STRIDE
#define VLG_ABL_AF_SEQ_STRIDE ((size_t)Q * V)
#endif
                f32x4* dst4 = reinterpret_cast<f32x4*>(out_full + ((((size_t)blockIdx.y * gridDim.x + blockIdx.x) * kAMWaves + wave) * a_per_block + i) * VLG_ABL_AF_SEQ_STRIDE);
#else
                f32x4* dst4 = reinterpret_cast<f32x4*>(out_full + (((size_t)bc * A + a) * Q + q0) * V);
#endif
                const f32x4* src4 = reinterpret_cast<const f32x4*>(otile);
                // (round 5) store instruction k covers the 16-byte pieces 64 k - sh + lane, sh = the chunk's first piece within its 128-byte line:
                // every instruction then writes eight WHOLE lines -- a chunk is Q V floats = 11 808 bytes at config-2, 32 bytes past a line
                // boundary, and with pieces 64 k + lane every one of the twelve instructions split two lines with its neighbour (measured on the
                // store stream alone: 181 -> 161 us with line-aligned chunks; tools/time_align_full_ablation.sh)
                const int sh = (int)((reinterpret_cast<uintptr_t>(dst4) >> 4) & 7);
                constexpr int NC = (RT * 16 * 48 / 4 + 63) / 64 + 1;   // V <= 48; + 1: the shifted last pieces
                // non-temporal: the 774 MB tensor is written once and read by a later kernel, never by this one (round 4:
                // 0.197 -> 0.185 ms unmasked, 0.201 -> 0.160 ms with masks on the same box).  In two halves: all 16-byte registers
                // in flight at once put the kernel over its 256 registers (the nt form keeps an address pair per store).
                constexpr int NH = (NC + 1) / 2;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    f32x4 tv[NH];
#pragma unroll
                    for (int k = 0; k < NH; ++k) tv[k] = src4[min(max(lane + 64 * (half * NH + k) - sh, 0), n4 - 1)];
#pragma unroll
                    for (int k = 0; k < NH; ++k) {
                        const int j = lane + 64 * (half * NH + k) - sh;
#ifdef VLG_ABL_AF_NOSTORE
                        if (half * NH + k < NC && j >= 0 && j < n4 && tv[k][0] == 1.2345e-33f) __builtin_nontemporal_store(tv[k], dst4 + j);
#elif defined(VLG_ABL_AF_PLAINST)
                        if (half * NH + k < NC && j >= 0 && j < n4) dst4[j] = tv[k];
#else
                        if (half * NH + k < NC && j >= 0 && j < n4) __builtin_nontemporal_store(tv[k], dst4 + j);
#endif
                    }
                }
            }
#ifdef VLG_ABL_AF_LDSBAR
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
            __syncthreads();   // (an LDS-only barrier -- s_waitcnt lgkmcnt(0) + s_barrier, leaving the stores in flight -- measured the same: 0.197 ms)
#endif
        }
    }
}

static int launch_align_full(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A, int Q, int V,
                             float neg_inf, float* out_full, hipStream_t s) {
    const int by = (B + kAMWaves - 1) / kAMWaves;
    int a_per_block = (int)(((long)A * by + 255) / 256);
    if (a_per_block < 8) a_per_block = 8;
    if (a_per_block > A) a_per_block = A;
#ifdef VLG_AF_APB   // (tools/ A/B builds)
    a_per_block = VLG_AF_APB;
#endif
    dim3 grid((A + a_per_block - 1) / a_per_block, by);
    const size_t lds = sizeof(float) * (size_t)kAMWaves * (VLG_AF_RT * 16) * V;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(align_full_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(align_full_kernel, grid, dim3(kAMThreads), lds, s, (const uint16_t*)txt, (const uint16_t*)vis, tmask, vmask, B, A, Q, V,
                       neg_inf, out_full, a_per_block);
    return check_launch("align_full_kernel");
}

// ---- float32 features -> two fp16 parts under one power-of-two scale per tensor (round 5; see align_argmax_kernel<NP = 2>) ----
// blockIdx.y = tensor (0: txt [n0 floats], 1: vis [n1 floats]).  First the 64 per-workgroup maxima of |x|, then the split: s = 2^k with
// |x|max s in [2^14, 2^15); an element 2^-j below the maximum keeps min(22, 39 - j) bits.
constexpr int kSplitBlocks = 64;
__global__ __launch_bounds__(256) void align_absmax_kernel(const float* __restrict__ x0, size_t n0, const float* __restrict__ x1, size_t n1,
                                                           float* __restrict__ part) {
    __shared__ float red[4];
    const float* x = blockIdx.y ? x1 : x0;
    const size_t n4 = (blockIdx.y ? n1 : n0) >> 2;
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)kSplitBlocks * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.y * kSplitBlocks + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ __launch_bounds__(256) void align_split_kernel(const float* __restrict__ x0, size_t n0, const float* __restrict__ x1, size_t n1,
                                                          const float* __restrict__ part, uint16_t* __restrict__ hi0, uint16_t* __restrict__ lo0,
                                                          uint16_t* __restrict__ hi1, uint16_t* __restrict__ lo1, float* __restrict__ inv) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    typedef __attribute__((ext_vector_type(2))) float f2;
    const int t = blockIdx.y;
    float m = part[t * kSplitBlocks + (threadIdx.x & 63)];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    int k = m > 0.f ? 14 - ((int)((__float_as_uint(m) >> 23) & 0xffu) - 127) : 0;
    k = k < -110 ? -110 : (k > 110 ? 110 : k);
    const float sc = __uint_as_float((uint32_t)(127 + k) << 23);
    if (blockIdx.x == 0 && threadIdx.x == 0) inv[t] = __uint_as_float((uint32_t)(127 - k) << 23);
    const float* x = t ? x1 : x0;
    uint16_t *hi = t ? hi1 : hi0, *lo = t ? lo1 : lo0;
    const size_t n4 = (t ? n1 : n0) >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        const f2 a = {v[0] * sc, v[1] * sc}, b = {v[2] * sc, v[3] * sc};
        const h2 ha = __builtin_convertvector(a, h2), hb = __builtin_convertvector(b, h2);
        const h2 la = __builtin_convertvector(a - __builtin_convertvector(ha, f2), h2), lb = __builtin_convertvector(b - __builtin_convertvector(hb, f2), h2);
        reinterpret_cast<uint2*>(hi)[i] = make_uint2(__builtin_bit_cast(uint32_t, ha), __builtin_bit_cast(uint32_t, hb));
        reinterpret_cast<uint2*>(lo)[i] = make_uint2(__builtin_bit_cast(uint32_t, la), __builtin_bit_cast(uint32_t, lb));
    }
}

// maxima + positions of float32 features (d = 128) on two fp16 parts: scratch = hi | lo of txt [B Q 128] and vis [A V 128] (uint16), then 128 partial
// maxima and the two inverse scales (floats): (B Q + A V) 128 floats + 256 floats in all
static int launch_align_argmax_f32(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A, int Q, int V,
                                   float neg_inf, float* out_maxV, float* out_maxQ, hipStream_t s, AlignArgs xa, float* scratch) {
    const size_t n0 = (size_t)B * Q * 128, n1 = (size_t)A * V * 128;
    uint16_t* th = reinterpret_cast<uint16_t*>(scratch);
    uint16_t *tl = th + n0, *vh = tl + n0, *vl = vh + n1;
    float* part = scratch + (n0 + n1);
    float* inv = part + 2 * kSplitBlocks;
    hipLaunchKernelGGL(align_absmax_kernel, dim3(kSplitBlocks, 2), dim3(256), 0, s, (const float*)txt, n0, (const float*)vis, n1, part);
    hipLaunchKernelGGL(align_split_kernel, dim3(256, 2), dim3(256), 0, s, (const float*)txt, n0, (const float*)vis, n1, part, th, tl, vh, vl, inv);
    if (int rc = check_launch("align_split_kernel")) return rc;
    const AlignParts parts{tl, vl, inv};
    const int by = (B + kAMWaves - 1) / kAMWaves;
    const int ng = (V + kAMRows - 1) / kAMRows;
    int a_per_block = (int)(((long)A * by + 255) / 256);
    if (a_per_block < (8 + ng - 1) / ng) a_per_block = (8 + ng - 1) / ng;
    if (a_per_block > A) a_per_block = A;
    dim3 grid((A + a_per_block - 1) / a_per_block, by);
#define VLG_AAM2(HQ, MU)                                                                                                      \
    hipLaunchKernelGGL((align_argmax_kernel<HQ, true, MU, 2>), grid, dim3(kAMThreads), 0, s, th, vh, tmask, vmask, B, A, Q, V, neg_inf, out_maxV, \
                       out_maxQ, a_per_block, xa, parts)
    if (out_maxQ) { if (ng > 1) VLG_AAM2(true, true); else VLG_AAM2(true, false); }
    else { if (ng > 1) VLG_AAM2(false, true); else VLG_AAM2(false, false); }
#undef VLG_AAM2
    if (int rc = check_launch("align_argmax_kernel (fp16 parts)")) return rc;
    if (xa.pen) {
        const int pd_nw = ng >= 4 ? 4 : 1;
        const size_t pd_lds = sizeof(float) * ((size_t)pd_nw * kPdRT * 16 * kPdP + (size_t)pd_nw * 256);
        hipError_t pe = hipFuncSetAttribute(reinterpret_cast<const void*>(align_prior_diag_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_lds);
        if (pe != hipSuccess) return set_error((int)pe, "hipFuncSetAttribute: %s", hipGetErrorString(pe));
        hipLaunchKernelGGL(align_prior_diag_kernel<2>, dim3(std::min(A, B)), dim3(64 * pd_nw), pd_lds, s, th, vh, tmask, vmask, B, A, Q, V, neg_inf, out_maxV,
                           out_maxQ, xa, parts);
        return check_launch("align_prior_diag_kernel (fp16 parts)");
    }
    return 0;
}

template <bool ARGS>
static int launch_align_max(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A, int Q,
                            int V, float neg_inf, float* out_maxV, float* out_maxQ, hipStream_t s,
                            AlignArgs xa = AlignArgs{nullptr, nullptr, 0, nullptr, nullptr}) {
    const int by = (B + kAMWaves - 1) / kAMWaves;
    // enough workgroups for the chip (256 CUs, one 8-wave workgroup each), but at least 8 image tiles per workgroup so that
    // the caption fragments are amortised (several region groups per image -- align_argmax_kernel only -- are several tiles)
    const int ng = ARGS ? (V + kAMRows - 1) / kAMRows : 1;
    int a_per_block = (int)(((long)A * by + 255) / 256);
    if (a_per_block < (8 + ng - 1) / ng) a_per_block = (8 + ng - 1) / ng;
    if (a_per_block > A) a_per_block = A;
    dim3 grid((A + a_per_block - 1) / a_per_block, by);   // x fastest: workgroups of one caption octet spread over the XCDs
    // positions wanted: both products on the matrix cores (172 vs 267 us at config-2).  The maxima alone stay with align_max_kernel:
    // without the searches that one is not vector-ALU bound, and the second product only costs (ARGS = false measured: 149 vs 130 us)
    if constexpr (ARGS) {
        if (!VLG_ENV("VLG_ALIGN_ARGMAX_OLD")) {   // (the env switch: tools/ A-B timing only)
            if (out_maxQ)
#define VLG_AAM(HQ, MU)                                                                                                       \
    hipLaunchKernelGGL((align_argmax_kernel<HQ, true, MU>), grid, dim3(kAMThreads), 0, s, (const uint16_t*)txt, (const uint16_t*)vis, tmask, \
                       vmask, B, A, Q, V, neg_inf, out_maxV, out_maxQ, a_per_block, xa, AlignParts{nullptr, nullptr, nullptr})
                { if (ng > 1) VLG_AAM(true, true); else VLG_AAM(true, false); }
            else
                { if (ng > 1) VLG_AAM(false, true); else VLG_AAM(false, false); }
#undef VLG_AAM
            if (xa.pen) {
                if (int rc = check_launch("align_argmax_kernel")) return rc;
                // wavefronts per diagonal pair: the region groups of a many-column image are dealt round.  Eight (round 6: 157 KB of score
                // tiles, the whole LDS of a CU) for the shipped layout's 29 groups: there are only B workgroups, the kernel is a chain of
                // dependent rounds, and 4 rounds instead of 8 took it from 162 to ~90 us at B = 64
                const int pd_nw = ng >= 8 ? 8 : ng >= 4 ? 4 : 1;
                const size_t pd_lds = sizeof(float) * ((size_t)pd_nw * kPdRT * 16 * kPdP + (size_t)pd_nw * 256);
                hipError_t pe = hipFuncSetAttribute(reinterpret_cast<const void*>(align_prior_diag_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pd_lds);
                if (pe != hipSuccess) return set_error((int)pe, "hipFuncSetAttribute: %s", hipGetErrorString(pe));
                hipLaunchKernelGGL(align_prior_diag_kernel<1>, dim3(std::min(A, B)), dim3(64 * pd_nw), pd_lds, s, (const uint16_t*)txt, (const uint16_t*)vis,
                                   tmask, vmask, B, A, Q, V, neg_inf, out_maxV, out_maxQ, xa, AlignParts{nullptr, nullptr, nullptr});
                return check_launch("align_prior_diag_kernel");
            }
            return check_launch("align_argmax_kernel");
        }
    }
    hipLaunchKernelGGL((align_max_kernel<ARGS, ARGS ? 3 : 6>), grid, dim3(kAMThreads), 0, s, (const uint16_t*)txt, (const uint16_t*)vis,
                       tmask, vmask, B, A, Q, V, neg_inf, out_maxV, out_maxQ, a_per_block, xa);
    return check_launch("align_max_kernel");
}

// =====================================================================================================
// Backward of the materialised alignment tensor (joint.py:413-418 under loss.backward()): given the cotangent
// g [B,A,Q,V] of attmap,
//   d_txt[b,q,:] = tmask[b,q] * sum_{a,v} g[b,a,q,v] * vmask[a,v] * vis[a,v,:]          (SIDE 0)
//   d_vis[a,v,:] = vmask[a,v] * sum_{b,q} g[b,a,q,v] * tmask[b,q] * txt[b,q,:]          (SIDE 1)
// (masked_fill_ passes no gradient).  Both are the same loop with different strides into g: an output row tile
// (16 rows of M = Q | V) per wavefront, all 128 feature columns in its accumulators, an outer loop over the O = A | B
// tensors of the other side and the contraction over K = V | Q inside it.  v_mfma_f32_16x16x4_f32: exact fp32
// products (the reference's numerics) and -- one element per lane per k -- no layout constraint, so g is read ONCE,
// in place, with both masks folded into the operand load: no permuted / masked / up-cast copies of the 774 MB
// cotangent (the torch.matmul formulation made three).  The other side's feature tile streams through a
// double-buffered LDS tile shared by the block's waves (pitch 144: the four k rows of a fragment read hit disjoint banks).
// =====================================================================================================
constexpr int kBwdThreads = 256;

// NCT = d / 16 column tiles; NS = k-steps (of 4) per contraction chunk, a compile-time count: a run-time bound on the
// unrolled step loop makes hipcc shuttle all 32 accumulator registers between AGPRs and VGPRs around every step
template <typename In, int NCT, int NS>
__global__ __launch_bounds__(kBwdThreads) void align_bwd_kernel(
    const float* __restrict__ g, const typename In::T* __restrict__ feat, const uint8_t* __restrict__ kmask,
    const uint8_t* __restrict__ rmask, int O, int M, int K, long so, long sr, long sk, long sfix, int o_per_block,
    float* __restrict__ out, int use_atomic) {
    using T = typename In::T;
    constexpr int D = NCT * 16, PITCH = D + 16, MAXS = NS, Kc = NS * 4;   // Kc: contraction chunk = rows of one LDS tile
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int nck = (K + Kc - 1) / Kc;                            // contraction chunks per outer index
    T* tile0 = reinterpret_cast<T*>(smem_raw);
    T* tile1 = tile0 + (size_t)Kc * PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gk = lane >> 4, ccol = lane & 15;
    const int fix = blockIdx.y, rt = blockIdx.x * 4 + wave;
    const int o0 = blockIdx.z * o_per_block, o1 = min(O, o0 + o_per_block);
    const int it0 = o0 * nck, it1 = o1 * nck;                     // iterations: (outer index, chunk) pairs
    const int row = rt * 16 + ccol, rowc = min(row, M - 1);      // A-operand row of this lane (clamped; rows >= M are never stored)
    const bool wave_live = rt * 16 < M;
    const float* gb = g + (size_t)fix * sfix + (size_t)rowc * sr;
    // the other side's feature tile of iteration `it`: global -> registers (issued one iteration ahead, so the loads land under
    // the MFMAs) -> LDS (after the MFMAs)
    constexpr int EPV = 16 / sizeof(T), VPR = D / EPV, NV = (Kc * VPR + kBwdThreads - 1) / kBwdThreads;
    auto stage_load = [&](int it, uint4* x) {
        const int o = it / nck, k0 = (it - o * nck) * Kc;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * kBwdThreads, k = i / VPR, c = (i - k * VPR) * EPV;
            x[j] = make_uint4(0, 0, 0, 0);                        // rows past K, and rows the contraction-side mask drops: zeros
            if (k < Kc && k0 + k < K && !(kmask && !kmask[(size_t)o * K + k0 + k]))
                x[j] = *reinterpret_cast<const uint4*>(feat + ((size_t)o * K + k0 + k) * D + c);
        }
    };
    auto stage_write = [&](T* tile, const uint4* x) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * kBwdThreads, k = i / VPR, c = (i - k * VPR) * EPV;
            if (k < Kc) *reinterpret_cast<uint4*>(tile + (size_t)k * PITCH + c) = x[j];
        }
    };
    auto load_a = [&](int it, float* a) {   // this lane's A elements: k = k0 + 4 t + gk, both masks' contraction side folded in
        const int o = it / nck, k0 = (it - o * nck) * Kc;
        const float* p = gb + (size_t)o * so;
        // unconditional loads on a clamped index (no branch between them: all in flight together).  The contraction-side mask
        // is NOT applied here: the masked feature rows are staged as zeros, which removes those terms just the same.
#pragma unroll
        for (int t = 0; t < MAXS; ++t) a[t] = p[(size_t)min(k0 + 4 * t + gk, K - 1) * sk];
#pragma unroll
        for (int t = 0; t < MAXS; ++t)
            if (k0 + 4 * t + gk >= K) a[t] = 0.f;
    };
    f32x4 acc[NCT];
#pragma unroll
    for (int c = 0; c < NCT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // The cotangent comes from HBM: its loads run two iterations ahead of their use, in three register sets whose roles
    // rotate by unrolling the iteration loop three times (copying a set into the next would wait for the loads just issued).
    float a0[MAXS], a1[MAXS], a2[MAXS];
    uint4 xs[NV];
    if (it0 < it1) {
        stage_load(it0, xs);
        stage_write(tile0, xs);
        if (wave_live) {
            load_a(it0, a0);
            if (it0 + 1 < it1) load_a(it0 + 1, a1);
        }
    }
    __syncthreads();
    auto body = [&](int it, const float* cur, float* ahead) {
        T* tcur = ((it - it0) & 1) ? tile1 : tile0;
        T* tnxt = ((it - it0) & 1) ? tile0 : tile1;
        if (it + 1 < it1) stage_load(it + 1, xs);
        if (wave_live && it + 2 < it1) load_a(it + 2, ahead);
        if (wave_live) {
            float bb[2][NCT];   // B fragments one k-step ahead of the MFMAs (the LDS latency hides under the previous step)
            const T* bbase = tcur + (size_t)gk * PITCH + ccol;
#pragma unroll
            for (int c = 0; c < NCT; ++c) bb[0][c] = In::ld(bbase, c * 16);
#pragma unroll
            for (int t = 0; t < MAXS; ++t) {
                if (t + 1 < MAXS) {
#pragma unroll
                    for (int c = 0; c < NCT; ++c) bb[(t + 1) & 1][c] = In::ld(bbase + (size_t)(4 * (t + 1)) * PITCH, c * 16);
                }
#pragma unroll
                for (int c = 0; c < NCT; ++c)
                    acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[t], bb[t & 1][c], acc[c], 0, 0, 0);
            }
        }
        if (it + 1 < it1) stage_write(tnxt, xs);
        __syncthreads();
    };
    for (int it = it0; it < it1; it += 3) {
        body(it, a0, a2);
        if (it + 1 < it1) body(it + 1, a1, a0);
        if (it + 2 < it1) body(it + 2, a2, a1);
    }
    if (!wave_live) return;
    // accumulator tile: lane l, register n <-> row 4 (l >> 4) + n, column l & 15
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int r = rt * 16 + gk * 4 + n;
        if (r >= M) continue;
        const float keep = (rmask && !rmask[(size_t)fix * M + r]) ? 0.f : 1.f;
        float* dst = out + ((size_t)fix * M + r) * D + ccol;
#pragma unroll
        for (int c = 0; c < NCT; ++c) {
            const float v = acc[c][n] * keep;
            if (use_atomic) atomicAdd(dst + c * 16, v);   // two partial sums per element at most: order-free
            else dst[c * 16] = v;
        }
    }
}

template <bool F32IN, int KCH, bool TILE, bool ARGS = false, int RTBV = MfmaCfg<F32IN>::RTB, bool DIRECT = false>
static int launch_align_mfma(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A,
                             int Q, int V, float neg_inf, float* out_full, float* out_maxV, float* out_maxQ,
                             float* out_diag, hipStream_t s, AlignArgs xa = AlignArgs{nullptr, nullptr, 0, nullptr, nullptr},
                             bool diag_only = false) {
    using C = MfmaCfg<F32IN>;
    int a_per_wave = diag_only ? -1 : A >= 2048 ? 16 : A >= 64 ? 8 : 1;
    // small batches (the shipped B = 64): fewer images per wave rather than fewer workgroups than CUs
    while (a_per_wave > 1 && (long)((A + 4 * a_per_wave - 1) / (4 * a_per_wave)) * B < 512) a_per_wave >>= 1;
    const int n_grp = (V + kCTB * 16 - 1) / (kCTB * 16);
    dim3 grid(diag_only ? std::max(1, std::min((n_grp + 3) / 4, 16)) : (A + 4 * a_per_wave - 1) / (4 * a_per_wave), B);
    constexpr int QB = RTBV * 16;
    const size_t lds = TILE ? sizeof(float) * 4 * (size_t)(QB * kTileVP + 2 * QB) : 0;
    auto k = align_mfma_kernel<F32IN, KCH, TILE, ARGS, RTBV, DIRECT>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, grid, dim3(kAlignThreads), lds, s, (const typename C::T*)txt, (const typename C::T*)vis, tmask,
                       vmask, B, A, Q, V, neg_inf, out_full, out_maxV, out_maxQ, out_diag, a_per_wave, xa);
    return check_launch("align_mfma_kernel");
}

// =====================================================================================================
// The same adjoint on the bf16 matrix cores (bf16 features, d = 128, rows and contraction length <= 96 per pair -- config-2).
// The fp32 MFMA above runs at 1/16 of the bf16 rate and made the kernel compute-bound (2.2 ms for a 774 MB cotangent).  Here
// the fp32 cotangent is split on the fly into NT bf16 terms, g = t0 + t1 (+ t2), t0 = bf16(g), t1 = bf16(g - t0), ...; the
// features are bf16 already, every product t_i * x is exact in the fp32 accumulator, and what is dropped is below
// 2^-17 |g| (NT = 2) / 2^-25 |g| (NT = 3: fp32's own rounding level).  The kernel then waits on HBM, not on the matrix cores.
//   A operand: the cotangent tile of one (fix, o) pair, straight from global memory in fragment order (a lane's 8 contraction
//     positions are contiguous for the caption side; 8 strided words for the image side), raw fp32 in registers for one
//     step, converted right before use;
//   B operand: the other side's features, contraction-major ([o][128][Kp] bf16 scratch, zero where the contraction mask is
//     off), one tile per step through LDS (double-buffered), shared by the block's four waves, which tile the [MT x 8]
//     output tiles RW x CW.
// One barrier per step: at the top of step o everything that was in flight (cotangent of o, features of o+1) is consumed --
// converted / written to the other LDS buffer -- and the next loads are issued before the MFMAs of step o start.
// =====================================================================================================
typedef __attribute__((ext_vector_type(8))) __bf16 ab_bf16x8;
typedef __attribute__((ext_vector_type(4))) float ab_f32x4;

// feat [O][K][128] -> featT [O][128][Kp] bf16, zero where mask[o][k] == 0 or k >= K: block = one 32-position strip.
// bf16 features are copied; fp32 features are split into two bf16 parts x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), the lo
// parts going to a second array of the same shape (`featT_lo`).
template <typename T>
__global__ __launch_bounds__(256) void align_bwd_transpose_kernel(const T* __restrict__ feat, const uint8_t* __restrict__ mask,
                                                                  int K, int Kp, uint16_t* __restrict__ featT,
                                                                  uint16_t* __restrict__ featT_lo) {
    constexpr bool F32 = sizeof(T) == 4;
    __shared__ uint16_t t[F32 ? 2 : 1][32][128 + 2];
    const int o = blockIdx.x, k0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int k = i >> 7, c = i & 127;
        const bool on = k0 + k < K && (!mask || mask[(size_t)o * K + k0 + k]);
        const T x = on ? feat[((size_t)o * K + k0 + k) * 128 + c] : (T)0;
        if constexpr (F32) {
            const __bf16 h = (__bf16)x;
            t[0][k][c] = __builtin_bit_cast(uint16_t, h);
            t[F32 ? 1 : 0][k][c] = __builtin_bit_cast(uint16_t, (__bf16)(x - (float)h));
        } else {
            t[0][k][c] = x;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int c = i >> 5, k = i & 31;
        featT[((size_t)o * 128 + c) * Kp + k0 + k] = t[0][k][c];
        if constexpr (F32) featT_lo[((size_t)o * 128 + c) * Kp + k0 + k] = t[F32 ? 1 : 0][k][c];
    }
}

// MT row tiles x CW column groups = 6 waves: each cotangent row tile is loaded by CW waves only (it is the HBM stream; the
// features sit in LDS, where re-reading them per wave is cheap)
// FS = feature parts: 1 = bf16 features; 2 = fp32 features as hi + lo bf16 tiles (featT holds the hi parts of all O tensors, then
// the lo parts): g x = g_hi x_hi + g_hi x_lo + g_lo x_hi, dropping g_lo x_lo (< 2^-16 of the product).
// NF = fixed indices (captions / images) per workgroup: their wave sets share every staged feature tile.  Round 3: per step a block
// moves 11.8 KB of cotangent from HBM and a 16-24 KB feature tile from L2; at the ~5.5 TB/s the HBM + L2 -> CU paths delivered
// together in every kernel of this shape, the RE-STREAMED features were more than half of what bounds it.  NF = 2 halves them.
// four consecutive cotangent words (4-byte aligned).  (Round 4: a NON-TEMPORAL load here -- the cotangent is read once per side --
// measured 0.53 -> 0.59 ms for the call: the second side's pass finds part of the 774 MB in the Infinity Cache only if the first
// side's loads were allowed to stay there.  -DVLG_BWD_NT selects it.)
typedef float ab_g4_t __attribute__((ext_vector_type(4), aligned(4)));
__device__ __forceinline__ float4 ld_g4(const char* p) {
#ifdef VLG_BWD_NT
    const ab_g4_t v = __builtin_nontemporal_load(reinterpret_cast<const ab_g4_t*>(p));
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}

template <bool KCONTIG, int NKC, int MT, int CW, int NT, int FS, int NF = 1>
__device__ __forceinline__ void align_bwd_split_body(const float* __restrict__ g, const uint16_t* __restrict__ featT,
                                                     const uint8_t* __restrict__ rmask, int O, int M, int K, long so, long sr,
                                                     long sk, long sfix, int o_per, float* __restrict__ out, int atomic, int nfix) {
    constexpr int Kp = NKC * 32, PITCH = Kp * 2 + 32, SEGS = Kp / 8;   // +32: conflict-free ds_read_b128 fragment reads
    static_assert(MT * CW == 6 || MT * CW == 3, "six waves, or three when every cotangent element is to be loaded once");
    constexpr int RT = 1, CT = 8 / CW, WPF = MT * CW, nthr = 64 * WPF * NF;
    constexpr int NV = (FS * 128 * SEGS + nthr - 1) / nthr, TILE = 128 * PITCH;   // LDS: [buffer][part][128][PITCH]
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % WPF, fi = (tid >> 6) / WPF, kg = lane >> 4, ccol = lane & 15;
    const int fix_raw = blockIdx.x * NF + fi, fix = min(fix_raw, nfix - 1);   // a wave set past the end mirrors the last index, stores nothing
    const int o_begin = blockIdx.y * o_per, o_end = min(O, o_begin + o_per);
    const int rt0 = (wave / CW) * RT, ct0 = (wave % CW) * CT;
    if (o_begin >= o_end) return;
    const float* gfix = g + (size_t)fix * sfix;
    // ---- raw loads (no arithmetic on the results until the next step: see ground_bwd_dense_kernel) ----
    uint4 xs[NV];
    float graw[RT][NKC][8];
#pragma unroll
    for (int j = 0; j < NV; ++j) xs[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int j = 0; j < 8; ++j) graw[r][kc][j] = 0.f;
    auto load_tile = [=](int o, uint4* xs) __attribute__((always_inline)) {
        const uint4* src = reinterpret_cast<const uint4*>(featT + (size_t)min(o, O - 1) * 128 * Kp);
        const size_t part_stride = (size_t)O * 128 * Kp / 8;   // uint4 between the hi and the lo array
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(tid + j * nthr, FS * 128 * SEGS - 1), part = i / (128 * SEGS), e = i - part * (128 * SEGS);
            xs[j] = src[part * part_stride + e];
        }
    };
    auto store_tile = [=](int buf, const uint4* xs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * nthr;
            if (i < FS * 128 * SEGS) {
                const int part = i / (128 * SEGS), e = i - part * (128 * SEGS), row = e / SEGS, seg = e - row * SEGS;
                *reinterpret_cast<uint4*>(smem_raw + (buf * FS + part) * TILE + row * PITCH + seg * 16) = xs[j];
            }
        }
    };
    // lane offsets into one pair's cotangent block, computed once (32-bit: a pair's block is small); per step only the
    // block-uniform base pointer moves, so a load costs no vector arithmetic
    const bool k_quads = (K & 3) == 0 && K >= 4;
    unsigned goff[RT][NKC][KCONTIG ? 2 : 8];   // BYTE offsets, unsigned: scalar base + zero-extended 32-bit vector offset is an addressing mode
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int row = min((rt0 + r) * 16 + ccol, M - 1);   // rows past M are computed on a copy of the last row and dropped
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            const int k0 = kc * 32 + kg * 8;
            if (KCONTIG) {
#pragma unroll
                for (int h = 0; h < 2; ++h) goff[r][kc][h] = 4u * (unsigned)(row * (int)sr + min(k0 + 4 * h, k_quads ? K - 4 : K - 1));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) goff[r][kc][KCONTIG ? 0 : j] = 4u * (unsigned)(row * (int)sr + min(k0 + j, K - 1) * (int)sk);
            }
        }
    }
    auto load_g = [&](int o, float (*graw)[NKC][8]) __attribute__((always_inline)) {
        const char* go = reinterpret_cast<const char*>(gfix + (size_t)min(o, O - 1) * so);   // block-uniform
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc) {
                if (KCONTIG) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (k_quads) {   // K % 4 == 0 (block-uniform): every quad is whole or empty; the empty ones re-read the
                                         // row's last quad and meet zero feature columns
                            const float4 v = ld_g4(go + goff[r][kc][h]);   // 4-byte aligned is enough
                            graw[r][kc][4 * h + 0] = v.x; graw[r][kc][4 * h + 1] = v.y;
                            graw[r][kc][4 * h + 2] = v.z; graw[r][kc][4 * h + 3] = v.w;
                        } else {         // clamped words
                            const int k0 = kc * 32 + kg * 8 + 4 * h;
                            const unsigned base = goff[r][kc][h] - 4u * (unsigned)min(k0, K - 1);
#pragma unroll
                            for (int j = 0; j < 4; ++j) graw[r][kc][4 * h + j] = *reinterpret_cast<const float*>(go + base + 4u * (unsigned)min(k0 + j, K - 1));
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) graw[r][kc][j] = *reinterpret_cast<const float*>(go + goff[r][kc][KCONTIG ? 0 : j]);
                }
            }
    };
    ab_f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = ab_f32x4{0.f, 0.f, 0.f, 0.f};
    // zero both tile buffers once (pitch padding is never written again), then the prologue: tile(o_begin) in place, tile
    // (o_begin + 1) and the cotangent of o_begin in flight
    for (int i = tid; i < 2 * FS * TILE / 16; i += nthr) reinterpret_cast<uint4*>(smem_raw)[i] = make_uint4(0, 0, 0, 0);
    load_tile(o_begin, xs);
    __syncthreads();
    store_tile(0, xs);
    load_g(o_begin, graw);
    load_tile(o_begin + 1, xs);
    __syncthreads();
    // one step: everything below is unconditional (indices clamped, a step past the end multiplies zeros), so that the compiler
    // can count the loads that stay in flight across the consumption of the older ones
    auto step = [&](int o, float (*gr)[NKC][8]) __attribute__((always_inline)) {
        const int buf = (o - o_begin) & 1;
        const bool real = o < o_end;
        // ---- consume: cotangent of o -> bf16 terms, features of o+1 -> the other buffer; reissue both ----
        ab_bf16x8 af[NT][RT][NKC];
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    // (positions past K hold a clamped, finite cotangent value and meet zero feature columns: no select)
                    float v = real ? gr[r][kc][j] : 0.f;
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const __bf16 h = (__bf16)v;
                        af[t][r][kc][j] = h;
                        v -= (float)h;
                    }
                }
        store_tile(buf ^ 1, xs);
        load_g(o + 1, gr);
        load_tile(o + 2, xs);
        // ---- MFMAs of pair o ----
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int part = 0; part < FS; ++part) {
                ab_bf16x8 bf[CT];
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    bf[c] = *reinterpret_cast<const ab_bf16x8*>(smem_raw + (buf * FS + part) * TILE + ((ct0 + c) * 16 + ccol) * PITCH +
                                                                 (kc * 4 + kg) * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < (part == 0 ? NT : 1); ++t)   // the lo feature part meets the leading cotangent term only
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < CT; ++c)
                            acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][r][kc], bf[c], acc[r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        __syncthreads();
    };
    // (a second cotangent set, loaded two steps ahead, was measured: no faster -- the step is bound by instruction issue, not
    //  by the latency of these loads -- and costs 16-24 VGPRs)
    for (int o = o_begin; o < o_end; ++o) step(o, graw);
    if (fix_raw >= nfix) return;   // (after the last barrier)
    // accumulator tile: lane l, register n <-> row 4 (l >> 4) + n, column l & 15
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (rt0 + r >= MT) break;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rr = (rt0 + r) * 16 + kg * 4 + n;
            if (rr < M) {
                const float keep = !rmask || rmask[(size_t)fix * M + rr] ? 1.f : 0.f;
                float* dst = out + ((size_t)fix * M + rr) * 128 + ct0 * 16 + ccol;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (atomic) atomicAdd(dst + c * 16, keep * acc[r][c][n]);
                    else dst[c * 16] = keep * acc[r][c][n];
                }
            }
        }
    }
}

// bf16 features: three waves per SIMD = two resident blocks per CU (the second block's work hides the first one's load latency)
template <bool KCONTIG, int NKC, int MT, int CW, int NT, int NF>
__global__ __launch_bounds__(64 * MT * CW * NF) __attribute__((amdgpu_waves_per_eu(2, 4))) void align_bwd_split_kernel(
    const float* __restrict__ g, const uint16_t* __restrict__ featT, const uint8_t* __restrict__ rmask, int O, int M, int K, long so,
    long sr, long sk, long sfix, int o_per, float* __restrict__ out, int atomic, int nfix) {
    align_bwd_split_body<KCONTIG, NKC, MT, CW, NT, 1, NF>(g, featT, rmask, O, M, K, so, sr, sk, sfix, o_per, out, atomic, nfix);
}

// fp32 features, split: twice the LDS per block (one block per CU) and more registers
template <bool KCONTIG, int NKC, int MT, int CW, int NT>
__global__ __launch_bounds__(384) void align_bwd_split_f32_kernel(
    const float* __restrict__ g, const uint16_t* __restrict__ featT, const uint8_t* __restrict__ rmask, int O, int M, int K, long so,
    long sr, long sk, long sfix, int o_per, float* __restrict__ out, int atomic, int nfix) {
    align_bwd_split_body<KCONTIG, NKC, MT, CW, NT, 2, 1>(g, featT, rmask, O, M, K, so, sr, sk, sfix, o_per, out, atomic, nfix);
}

// Caption side with a short, quad-aligned contraction (K % 4 == 0, 2 K <= 96 -- config-2's 36 regions): TWO pairs per step,
// their contraction ranges side by side in one 96-wide fragment set (72 of 96 positions live instead of 36 of 64).  The kernel
// above is bound by the bytes it keeps in flight (a wave waits a full loaded-HBM latency per step for 2.3 KB of cotangent);
// this doubles them and drops a quarter of the MFMAs.  The features come from a scratch that is concatenated along the
// contraction axis over ALL outer indices, featC [128][O K (+ padding)], so that a step's tile is 16-byte-aligned rows of it.
__global__ __launch_bounds__(256) void align_bwd_concat_transpose_kernel(const uint16_t* __restrict__ feat, const uint8_t* __restrict__ mask,
                                                                         int OK, long pitch, uint16_t* __restrict__ featC) {
    __shared__ uint16_t t[32][128 + 2];
    const int k0 = blockIdx.x * 32;   // position in the concatenated axis (o K + k)
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int k = i >> 7, c = i & 127;
        const bool on = k0 + k < OK && (!mask || mask[k0 + k]);
        t[k][c] = on ? feat[(size_t)(k0 + k) * 128 + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int c = i >> 5, k = i & 31;
        if (k0 + k < pitch) featC[(size_t)c * pitch + k0 + k] = t[k][c];
    }
}

// NF captions per workgroup (six waves each) share every staged tile, as in align_bwd_split_body
template <int MT, int CW, int NT, int NF>
__global__ __launch_bounds__(384 * NF) __attribute__((amdgpu_waves_per_eu(3, 4))) void align_bwd_split2_kernel(
    const float* __restrict__ g, const uint16_t* __restrict__ featC, long pitchC, const uint8_t* __restrict__ rmask, int O, int M,
    int K, long so, long sr, long sfix, int o_per, float* __restrict__ out, int atomic, int nfix) {
    constexpr int NKC = 3, Kp = 96, PITCH = Kp * 2 + 32, SEGS = Kp / 8;
    static_assert(MT * CW == 6, "six waves");
    constexpr int CT = 8 / CW, nthr = 384 * NF, NV = (128 * SEGS + nthr - 1) / nthr;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) % 6, fi = (tid >> 6) / 6, kg = lane >> 4, ccol = lane & 15;
    const int fix_raw = blockIdx.x * NF + fi, fix = min(fix_raw, nfix - 1);   // a wave set past the end mirrors the last caption, stores nothing
    const int o_begin = blockIdx.y * o_per, o_end = min(O, o_begin + o_per);   // o_per is even
    const int rt = wave / CW, ct0 = (wave % CW) * CT;
    if (o_begin >= o_end) return;
    const float* gfix = g + (size_t)fix * sfix;
    const int row = min(rt * 16 + ccol, M - 1);   // rows past M are computed on a copy of the last row and dropped
    uint4 xs[NV];
    float graw[NKC][8];
#pragma unroll
    for (int j = 0; j < NV; ++j) xs[j] = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int j = 0; j < 8; ++j) graw[kc][j] = 0.f;
    auto load_tile = [=](int o, uint4* xs) __attribute__((always_inline)) {   // columns o K ... o K + 95 of every feature row
        const uint16_t* src = featC + (size_t)min(o, O - 1) * K;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(tid + j * nthr, 128 * SEGS - 1), r = i / SEGS, seg = i - r * SEGS;
            xs[j] = *reinterpret_cast<const uint4*>(src + (size_t)r * pitchC + seg * 8);
        }
    };
    auto store_tile = [=](int buf, const uint4* xs) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = tid + j * nthr;
            if (i < 128 * SEGS) {
                const int r = i / SEGS, seg = i - r * SEGS;
                *reinterpret_cast<uint4*>(smem_raw + (buf * 128 + r) * PITCH + seg * 16) = xs[j];
            }
        }
    };
    // quad q of the concatenated axis (positions 4q .. 4q+3) lies in pair 4q / K at column 4q % K; quads past 2 K re-read
    // the first one and are zeroed at conversion.  Byte offsets from the step's (block-uniform) base pointer are computed once;
    // the second pair's quads add one pair stride (zero when the range ends on an odd pair).
    unsigned goff[NKC][2];
    bool gsec[NKC][2];
#pragma unroll
    for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kk = kc * 32 + kg * 8 + 4 * h;
            gsec[kc][h] = kk >= K && kk < 2 * K;
            goff[kc][h] = 4u * (unsigned)(row * (int)sr + (kk < 2 * K ? kk - (kk >= K ? K : 0) : 0));
        }
    auto load_g = [&](int o, float (*graw)[8]) __attribute__((always_inline)) {
        const char* go = reinterpret_cast<const char*>(gfix + (size_t)min(o, O - 1) * so);
        const unsigned step2 = o + 1 < O ? 4u * (unsigned)so : 0u;
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 q = ld_g4(go + goff[kc][h] + (gsec[kc][h] ? step2 : 0u));   // 4-byte aligned is enough
                graw[kc][4 * h + 0] = q.x; graw[kc][4 * h + 1] = q.y; graw[kc][4 * h + 2] = q.z; graw[kc][4 * h + 3] = q.w;
            }
    };
    ab_f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = ab_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 2 * 128 * PITCH / 16; i += nthr) reinterpret_cast<uint4*>(smem_raw)[i] = make_uint4(0, 0, 0, 0);
    load_tile(o_begin, xs);
    __syncthreads();
    store_tile(0, xs);
    load_g(o_begin, graw);
    load_tile(o_begin + 2, xs);
    __syncthreads();
    for (int o = o_begin; o < o_end; o += 2) {
        const int buf = ((o - o_begin) >> 1) & 1;
        const bool second = o + 1 < o_end;
        // ---- consume: cotangents of (o, o+1) -> bf16 terms, features of the next step -> the other buffer; reissue both ----
        ab_bf16x8 af[NT][NKC];
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = kc * 32 + kg * 8 + j;
                float v = kk < K || (kk < 2 * K && second) ? graw[kc][j] : 0.f;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const __bf16 h = (__bf16)v;
                    af[t][kc][j] = h;
                    v -= (float)h;
                }
            }
        store_tile(buf ^ 1, xs);
        load_g(o + 2, graw);
        load_tile(o + 4, xs);
#pragma unroll
        for (int kc = 0; kc < NKC; ++kc) {
            ab_bf16x8 bf[CT];
#pragma unroll
            for (int c = 0; c < CT; ++c)
                bf[c] = *reinterpret_cast<const ab_bf16x8*>(smem_raw + (buf * 128 + (ct0 + c) * 16 + ccol) * PITCH + (kc * 4 + kg) * 16);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int c = 0; c < CT; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][kc], bf[c], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    if (fix_raw >= nfix) return;   // (after the last barrier)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int rr = rt * 16 + kg * 4 + n;
        if (rr < M) {
            const float keep = !rmask || rmask[(size_t)fix * M + rr] ? 1.f : 0.f;
            float* dst = out + ((size_t)fix * M + rr) * 128 + ct0 * 16 + ccol;
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                if (atomic) atomicAdd(dst + c * 16, keep * acc[c][n]);
                else dst[c * 16] = keep * acc[c][n];
            }
        }
    }
}

}  // namespace vlg

extern "C" {

int vlg_bilinear_align(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A,
                       int Q, int V, int d, int in_dtype, float neg_inf, float* out_full, float* out_maxV,
                       float* out_maxQ, float* out_diag, void* stream) {
    using namespace vlg;
    if (B < 0 || A < 0 || Q < 1 || V < 1 || d < 1)
        return set_error(VLG_ERR_SHAPE, "bilinear_align: bad shape B=%d A=%d Q=%d V=%d d=%d", B, A, Q, V, d);
    if (out_diag && A != B) return set_error(VLG_ERR_SHAPE, "bilinear_align: out_diag needs A == B (A=%d B=%d)", A, B);
    if (B == 0 || A == 0) return 0;
    if (!txt || !vis) return set_error(VLG_ERR_ARG, "bilinear_align: null input");
    if (!out_full && !out_maxV && !out_maxQ && !out_diag) return set_error(VLG_ERR_ARG, "bilinear_align: no output requested");
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "bilinear_align: B=%d exceeds grid.y", B);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "bilinear_align: in_dtype %d", in_dtype);

    // ---- matrix-core path: d a whole number of MFMA K-chunks (the model's d = 128; also 64) ----
    const bool f32in = in_dtype == VLG_F32;
    const bool tile = out_full || out_diag || out_maxV;   // max over Q alone needs no LDS round trip
    // the full tensor is bound by its writes and keeps 96-row passes (bf16).  Fused maxima without the full tensor:
    //   * one region group per image (V <= 48, config-2): the LDS-tile path with 48-row passes (two blocks per CU: 0.274 ->
    //     0.257 ms); keeping max over V in registers instead measured the same or worse there (0.24 ms alone, 0.32 vs 0.28 ms
    //     with the diagonal block);
    //   * several groups per image (the shipped factor layout, V = 1369): max over V stays in registers across the groups
    //     and is folded once per image (DIRECT, no LDS: 1.28 -> 0.74 ms at B = 64), the diagonal block comes from a second
    //     launch over the B diagonal pairs;
    //   * the diagonal block alone: only those B pairs are computed (0.187 -> 0.012 ms).
#define VLG_MFMA(F32, KCHV)                                                                                         \
    if (tile && !out_full) {                                                                                        \
        if ((out_maxV || out_maxQ) && V <= kCTB * 16)                                                               \
            return launch_align_mfma<F32, KCHV, true, false, 3>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, nullptr, out_maxV, \
                                                                out_maxQ, out_diag, s);                             \
        if (out_maxV || out_maxQ) {                                                                                 \
            const int rc = launch_align_mfma<F32, KCHV, false, false, MfmaCfg<F32>::RTB, true>(                     \
                txt, vis, tmask, vmask, B, A, Q, V, neg_inf, nullptr, out_maxV, out_maxQ, nullptr, s);              \
            if (rc || !out_diag) return rc;                                                                         \
        }                                                                                                           \
        return launch_align_mfma<F32, KCHV, true, false, 3>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, nullptr, nullptr,  \
                                                            nullptr, out_diag, s, AlignArgs{nullptr, nullptr, 0, nullptr, nullptr}, true); \
    }                                                                                                               \
    return tile ? launch_align_mfma<F32, KCHV, true>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, out_full, out_maxV, \
                                                     out_maxQ, out_diag, s)                                           \
                : launch_align_mfma<F32, KCHV, false>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, out_full, out_maxV, \
                                                      out_maxQ, out_diag, s)
    if (!f32in && d == 128 && !out_full && (out_maxV || out_maxQ) && V <= kAMRows) {
        // fused maxima of one-group images: image tiles shared by eight captions through LDS
        if (int rc = launch_align_max<false>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, out_maxV, out_maxQ, s)) return rc;
        if (!out_diag) return 0;
        return launch_align_mfma<false, 4, true, false, 3>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, nullptr, nullptr, nullptr,
                                                          out_diag, s, AlignArgs{nullptr, nullptr, 0, nullptr, nullptr}, true);
    }
    // (eight waves' output blocks of 96 x V floats next to the 24 KB image tiles: V <= 44 fits the 160 KB LDS)
    if (!f32in && d == 128 && out_full && !out_maxV && !out_maxQ && !out_diag && V <= 44 && V % 4 == 0 && !VLG_ENV("VLG_ALIGN_FULL_OLD"))
        return launch_align_full(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, out_full, s);   // stores straight from the accumulators
    if (!f32in && d == 128) { VLG_MFMA(false, 4); }
    if (!f32in && d == 64) { VLG_MFMA(false, 2); }
    if (f32in && d == 128) { VLG_MFMA(true, 8); }
    if (f32in && d == 64) { VLG_MFMA(true, 4); }
#undef VLG_MFMA

    // ---- generic fp32-FMA path ----
    const size_t lds = sizeof(float) * ((size_t)(kQT + kVT) * (d + 1) + (size_t)kQT * (kVT + 1) + Q + V);
    if (lds > 160 * 1024) return set_error(VLG_ERR_SHAPE, "bilinear_align: d=%d Q=%d V=%d exceed the LDS tile budget", d, Q, V);
    int a_per_block = 8;
    dim3 grid((A + a_per_block - 1) / a_per_block, B);
#define VLG_LAUNCH(INV)                                                                                            \
    do {                                                                                                           \
        auto k = align_kernel<INV>;                                                                                \
        if (lds > 64 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));        \
        }                                                                                                          \
        hipLaunchKernelGGL(k, grid, dim3(kAlignThreads), lds, s, (const INV::T*)txt, (const INV::T*)vis, tmask,    \
                           vmask, B, A, Q, V, d, neg_inf, out_full, out_maxV, out_maxQ, out_diag, a_per_block);    \
    } while (0)
    if (f32in) VLG_LAUNCH(F32In);
    else VLG_LAUNCH(BF16In);
#undef VLG_LAUNCH
    return check_launch("align_kernel");
}

// bf16 features, d = 128, at most 96 rows and 96 contraction positions per pair: the split-term path on the bf16 matrix cores
static bool bwd_split_ok(int in_dtype, int d, int M, int K) {
    return (in_dtype == VLG_BF16 || (in_dtype == VLG_F32 && !VLG_ENV("VLG_BWD_F32_EXACT"))) && d == 128 && M <= 96 && K <= 96;
}

// caption side, two pairs per step on one concatenated scratch: short quad-aligned contraction
static bool bwd_concat_ok(int K) { return (K & 3) == 0 && K >= 4 && 2 * K <= 96 && !VLG_ENV("VLG_BWD_NOCONCAT"); }
static size_t bwd_concat_pitch(int O, int K) { return ((size_t)O * K + 96 + 7) / 8 * 8; }

size_t vlg_bilinear_align_backward_workspace(int B, int A, int Q, int V, int d, int in_dtype) {
    if (B < 1 || A < 1 || Q < 1 || V < 1) return 0;
    size_t n = 0;
    const size_t parts = in_dtype == VLG_F32 ? 2 : 1;   // fp32 features: hi and lo bf16 copies
    if (bwd_split_ok(in_dtype, d, Q, V))   // caption side: vis, contraction-major (concatenated over the images when V is short)
        n += bwd_concat_ok(V) && parts == 1 ? (size_t)128 * bwd_concat_pitch(A, V) : parts * A * 128 * ((V + 31) / 32 * 32);
    if (bwd_split_ok(in_dtype, d, V, Q)) n += parts * B * 128 * ((Q + 31) / 32 * 32);   // image side: txt
    return n * 2;
}

int vlg_bilinear_align_backward(const float* grad_out, const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask,
                                int B, int A, int Q, int V, int d, int in_dtype, void* ws, size_t ws_bytes, float* grad_txt,
                                float* grad_vis, void* stream) {
    using namespace vlg;
    if (B < 1 || A < 1 || Q < 1 || V < 1) return set_error(VLG_ERR_SHAPE, "bilinear_align_backward: bad shape B=%d A=%d Q=%d V=%d", B, A, Q, V);
    if (d != 128 && d != 64 && d != 32) return set_error(VLG_ERR_SHAPE, "bilinear_align_backward: d=%d (supported: 32, 64, 128)", d);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "bilinear_align_backward: in_dtype %d", in_dtype);
    if (!grad_out || !txt || !vis || (!grad_txt && !grad_vis)) return set_error(VLG_ERR_ARG, "bilinear_align_backward: null buffer");
    if (B > 65535 || A > 65535) return set_error(VLG_ERR_SHAPE, "bilinear_align_backward: B=%d A=%d exceed grid.y", B, A);
    const size_t need = vlg_bilinear_align_backward_workspace(B, A, Q, V, d, in_dtype);
    if (need && (!ws || ws_bytes < need))
        return set_error(VLG_ERR_WORKSPACE, "bilinear_align_backward: workspace %zu bytes, need %zu (vlg_bilinear_align_backward_workspace)",
                         ws_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = in_dtype == VLG_F32 ? 4 : 2;
    // ---- split-term path: the cotangent as bf16 terms against contraction-major bf16 features ----
    // terms of the fp32 -> bf16 split of the cotangent: two keep 16 significant bits (relative error < 2^-17 per product, far
    // below the bf16 rounding the features -- and the gradients autograd hands back to bf16 leaves -- already carry); a third
    // term (fp32's own rounding level) costs half as many MFMAs and conversions again: 0.56 vs 0.45 ms on the caption side
    constexpr int kNT = 2;
    const bool f32feat = in_dtype == VLG_F32;
    constexpr int kCW3 = 2;   // column groups for <= 48 rows (1 = three-wave blocks that load every cotangent element once: 2x slower, too few waves per CU)
    auto go_split = [&](const void* feat, const uint8_t* km, const uint8_t* rm, int fixn, int O, int M, int K, long so, long sr,
                        long sk, long sfix, bool kcontig, uint16_t* featT, float* out) -> int {
        const int Kp = (K + 31) / 32 * 32, nkc = Kp / 32;
        if (f32feat)
            hipLaunchKernelGGL(align_bwd_transpose_kernel<float>, dim3(O, Kp / 32), dim3(256), 0, s, (const float*)feat, km, K, Kp, featT,
                               featT + (size_t)O * 128 * Kp);
        else
            hipLaunchKernelGGL(align_bwd_transpose_kernel<uint16_t>, dim3(O, Kp / 32), dim3(256), 0, s, (const uint16_t*)feat, km, K, Kp,
                               featT, (uint16_t*)nullptr);
        // bf16 features: two fixed indices per workgroup share each staged feature tile (12 waves, one block per CU)
        const int nf = (!f32feat && fixn >= 64 && !VLG_ENV("VLG_BWD_NF1")) ? 2 : 1;
        int split = 1;   // the chip covered at least once; two-addend atomics are order-free
        if ((long)fixn * 2 <= 1024 * nf && O >= 16) split = 2;
        if (const char* e = VLG_ENV("VLG_BWD_SPLIT")) split = atoi(e) >= 2 ? 2 : 1;
        const int opb = (O + split - 1) / split;
        if (split > 1) {
            hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)fixn * M * 128, s);
            if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        const size_t lds = (f32feat ? 4 : 2) * (size_t)128 * (Kp * 2 + 32);
#define VLG_BS(KC, NKCV, MTV, CWV)                                                                                      \
        do {                                                                                                            \
            /* the two-index form where it fits the 168 registers of three waves per SIMD (12-wave blocks) */           \
            constexpr int NFV = (NKCV == 3 && (MTV == 6 || KC)) ? 1 : 2;                                                \
            const int nfe = nf == 2 ? NFV : 1;                                                                          \
            auto k = f32feat ? align_bwd_split_f32_kernel<KC, NKCV, MTV, CWV, kNT>                                      \
                             : (nfe == 2 ? align_bwd_split_kernel<KC, NKCV, MTV, CWV, kNT, NFV> : align_bwd_split_kernel<KC, NKCV, MTV, CWV, kNT, 1>); \
            if (lds > 64 * 1024) {                                                                                      \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));         \
            }                                                                                                           \
            hipLaunchKernelGGL(k, dim3((fixn + nfe - 1) / nfe, split), dim3(64 * MTV * CWV * nfe), lds, s, grad_out, featT, rm, O, M, K, so, \
                               sr, sk, sfix, opb, out, split > 1 ? 1 : 0, fixn);                                        \
        } while (0)
#define VLG_BS2(KC, NKCV)                                                                                               \
        do { if (M <= 48) VLG_BS(KC, NKCV, 3, kCW3); else VLG_BS(KC, NKCV, 6, 1); } while (0)
#define VLG_BS3(KC)                                                                                                     \
        do { if (nkc == 1) VLG_BS2(KC, 1); else if (nkc == 2) VLG_BS2(KC, 2); else VLG_BS2(KC, 3); } while (0)
        if (kcontig) VLG_BS3(true); else VLG_BS3(false);
#undef VLG_BS3
#undef VLG_BS2
#undef VLG_BS
        return check_launch("align_bwd_split_kernel");
    };
    auto go = [&](const void* feat, const uint8_t* km, const uint8_t* rm, int fixn, int O, int M, int K, long so, long sr, long sk,
                  long sfix, float* out) -> int {
        // k-steps per chunk from a small compile-time set (extra steps multiply zeros): 9 <-> V = 36, 21 <-> Q = 82
        const int steps = (K + 3) / 4;
        const int ns = steps <= 6 ? 6 : steps <= 9 ? 9 : steps <= 12 ? 12 : steps <= 21 ? 21 : 24;
        const size_t lds = 2 * (size_t)ns * 4 * (d + 16) * esz;
        const int tiles = (M + 15) / 16, gx = (tiles + 3) / 4;
        // split the outer range over several workgroups; their partial sums meet in a zeroed output by atomicAdd
        // more workgroups per CU hide the cotangent's HBM latency, but the partial sums meet by atomicAdd and only TWO addends are
        // order-free (a + b == b + a; three or more are not associative in fp32): the split stops at 2 -- bit-reproducible
        int split = 1;
        if ((long)gx * fixn < 768 && O / 2 >= 16) split = 2;
        if (const char* e = VLG_ENV("VLG_BWD_SPLIT")) split = atoi(e) >= 2 ? 2 : 1;   // tools/ experiments only
        const int opb = (O + split - 1) / split;
        if (split > 1) {
            hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)fixn * M * d, s);
            if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        dim3 grid(gx, fixn, split);
#define VLG_BWD3(INV, NCTV, NSV)                                                                                       \
        do {                                                                                                           \
            auto k = align_bwd_kernel<INV, NCTV, NSV>;                                                                 \
            if (lds > 64 * 1024) {                                                                                     \
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));        \
            }                                                                                                          \
            hipLaunchKernelGGL(k, grid, dim3(kBwdThreads), lds, s, grad_out, (const INV::T*)feat, km, rm, O, M, K, so, sr, sk,    \
                               sfix, opb, out, split > 1 ? 1 : 0);                                                     \
        } while (0)
#define VLG_BWD(INV, NCTV)                                                                                             \
        do {                                                                                                           \
            if (ns == 6) VLG_BWD3(INV, NCTV, 6); else if (ns == 9) VLG_BWD3(INV, NCTV, 9); else if (ns == 12) VLG_BWD3(INV, NCTV, 12); \
            else if (ns == 21) VLG_BWD3(INV, NCTV, 21); else VLG_BWD3(INV, NCTV, 24);                                  \
        } while (0)
        if (in_dtype == VLG_F32) { if (d == 128) VLG_BWD(F32In, 8); else if (d == 64) VLG_BWD(F32In, 4); else VLG_BWD(F32In, 2); }
        else { if (d == 128) VLG_BWD(BF16In, 8); else if (d == 64) VLG_BWD(BF16In, 4); else VLG_BWD(BF16In, 2); }
#undef VLG_BWD
#undef VLG_BWD3
        return check_launch("align_bwd_kernel");
    };
    const long QV = (long)Q * V;
    uint16_t* visT = reinterpret_cast<uint16_t*>(ws);
    const bool concat = bwd_split_ok(in_dtype, d, Q, V) && bwd_concat_ok(V) && !f32feat;
    uint16_t* txtT = visT + (!bwd_split_ok(in_dtype, d, Q, V) ? 0 : concat ? (size_t)128 * bwd_concat_pitch(A, V)
                                                                            : (f32feat ? 2 : 1) * (size_t)A * 128 * ((V + 31) / 32 * 32));
    if (grad_txt && concat) {   // two images per step, their 2 V <= 96 positions side by side
        const long pitch = (long)bwd_concat_pitch(A, V);
        hipLaunchKernelGGL(align_bwd_concat_transpose_kernel, dim3((unsigned)((pitch + 31) / 32)), dim3(256), 0, s, (const uint16_t*)vis,
                           vmask, A * V, pitch, visT);
        const int nf = (B >= 64 && !VLG_ENV("VLG_BWD_NF1")) ? 2 : 1;
        int split = ((long)B * 2 <= 1024 * nf && A >= 16) ? 2 : 1;
        if (const char* e = VLG_ENV("VLG_BWD_SPLIT")) split = atoi(e) >= 2 ? 2 : 1;
        const int opb = ((A + split - 1) / split + 1) & ~1;   // even: a step's tile rows start on 16-byte boundaries of featC
        if (split > 1) {
            hipError_t e = hipMemsetAsync(grad_txt, 0, sizeof(float) * (size_t)B * Q * 128, s);
            if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        const size_t lds = 2 * (size_t)128 * (96 * 2 + 32);
#define VLG_BS2K(MTV, CWV, NFV)                                                                                         \
        hipLaunchKernelGGL((align_bwd_split2_kernel<MTV, CWV, kNT, NFV>), dim3((B + NFV - 1) / NFV, split), dim3(384 * NFV), lds, s, grad_out, \
                           visT, pitch, tmask, A, Q, V, QV, (long)V, (long)A * QV, opb, grad_txt, split > 1 ? 1 : 0, B)
        if (Q <= 48) { if (nf == 2) VLG_BS2K(3, 2, 2); else VLG_BS2K(3, 2, 1); }
        else { if (nf == 2) VLG_BS2K(6, 1, 2); else VLG_BS2K(6, 1, 1); }
#undef VLG_BS2K
        if (int rc = check_launch("align_bwd_split2_kernel")) return rc;
    } else if (grad_txt) {   // rows q of caption b; outer a, contraction v:  g[((b A + a) Q + q) V + v]
        const int rc = bwd_split_ok(in_dtype, d, Q, V) ? go_split(vis, vmask, tmask, B, A, Q, V, QV, V, 1, (long)A * QV, true, visT, grad_txt)
                                                       : go(vis, vmask, tmask, B, A, Q, V, QV, V, 1, (long)A * QV, grad_txt);
        if (rc) return rc;
    }
    if (grad_vis) {   // rows v of image a; outer b, contraction q
        const int rc = bwd_split_ok(in_dtype, d, V, Q) ? go_split(txt, tmask, vmask, A, B, V, Q, (long)A * QV, 1, V, QV, false, txtT, grad_vis)
                                                       : go(txt, tmask, vmask, A, B, V, Q, (long)A * QV, 1, V, QV, grad_vis);
        if (rc) return rc;
    }
    return 0;
}


size_t vlg_grounding_loss_workspace(int B, int Q, int V) {
    if (B < 1 || Q < 1 || V < 1) return 0;
    return vlg::GroundPlan(B, Q, V).bytes;
}

int vlg_grounding_loss(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marginal,
                       const float* pen, const uint8_t* seg_of_v, int n_seg, int B, int Q, int V, int d, int in_dtype,
                       float neg_inf, float num_token, float w_vis2txt, void* ws, size_t ws_bytes, float* out_sums,
                       float* g_txt, float* g_vis, void* stream) {
    using namespace vlg;
    if (B < 1 || Q < 1 || V < 1 || d < 1)
        return set_error(VLG_ERR_SHAPE, "grounding_loss: bad shape B=%d Q=%d V=%d d=%d", B, Q, V, d);
    if (Q > 65535 || V > 32767) return set_error(VLG_ERR_SHAPE, "grounding_loss: Q=%d V=%d exceed the 16-bit position range", Q, V);
    if (d > 256) return set_error(VLG_ERR_SHAPE, "grounding_loss: d=%d > 256", d);
    if (!txt || !vis || !marginal || !out_sums) return set_error(VLG_ERR_ARG, "grounding_loss: null buffer");
    if (pen && (!seg_of_v || n_seg < 1)) return set_error(VLG_ERR_ARG, "grounding_loss: prior table without segments");
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "grounding_loss: B=%d exceeds grid.y", B);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "grounding_loss: in_dtype %d", in_dtype);
    const GroundPlan p(B, Q, V);
    if (!ws || ws_bytes < p.bytes) return set_error(VLG_ERR_WORKSPACE, "grounding_loss: workspace %zu bytes < %zu", ws_bytes, p.bytes);
    hipStream_t s = (hipStream_t)stream;
    float* wsf = (float*)ws;
    AlignArgs xa{pen, seg_of_v, n_seg, reinterpret_cast<uint16_t*>(wsf + p.off_argV), reinterpret_cast<uint16_t*>(wsf + p.off_argQ)};
    const bool f32in = in_dtype == VLG_F32;
    int rc = -1;
    // 48 query rows per pass for both dtypes: the per-wave LDS tile limits occupancy, and a second block per CU hides this
    // variant's long epilogue behind the other block's MFMAs (bf16, measured: 96 rows 535 us, 64 rows ~700, 48 rows 383,
    // 32 rows ~450; the plain tile paths gain nothing from it: 0.27 vs 0.25 ms full tensor, 0.26 vs 0.27 ms fused maxima)
#define VLG_GA_RTB 3
    // several region groups per image (V > 48, the shipped factor layout): (max, position) pairs stay in registers across
    // the groups instead of going through the LDS tile per group
#define VLG_GA(F32, KCHV)                                                                                          \
    rc = V > kCTB * 16                                                                                             \
             ? launch_align_mfma<F32, KCHV, false, true, VLG_GA_RTB, true>(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, nullptr,        \
                                                                         wsf + p.off_maxV, wsf + p.off_maxQ, nullptr, s, xa)        \
             : launch_align_mfma<F32, KCHV, true, true, VLG_GA_RTB>(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, nullptr,               \
                                                                    wsf + p.off_maxV, wsf + p.off_maxQ, nullptr, s, xa)
    // shared image tiles: align_argmax_kernel, any number of region groups (VLG_ALIGN_ARGMAX_OLD: the round-2 kernel for V <= 48)
    if (!f32in && d == 128 && (V <= kAMRows || !VLG_ENV("VLG_ALIGN_ARGMAX_OLD")) && !VLG_ENV("VLG_GROUND_OLD_ALIGN"))
        rc = launch_align_max<true>(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, wsf + p.off_maxV, wsf + p.off_maxQ, s, xa);
    else if (!f32in && d == 128) VLG_GA(false, 4);
    else if (!f32in && d == 64) VLG_GA(false, 2);
    else if (!f32in && d == 32) VLG_GA(false, 1);
    else if (f32in && d == 128 && Q <= 65535 && !VLG_ENV("VLG_ALIGN_F32_EXACT"))   // two fp16 parts per feature on the shared-image-tile kernel
        rc = launch_align_argmax_f32(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, wsf + p.off_maxV, wsf + p.off_maxQ, s, xa, wsf + p.off_parts);
    else if (f32in && d == 128) VLG_GA(true, 8);
    else if (f32in && d == 64) VLG_GA(true, 4);
    else if (f32in && d == 32) VLG_GA(true, 2);
    else return set_error(VLG_ERR_SHAPE, "grounding_loss: d=%d (supported: 32, 64, 128)", d);
#undef VLG_GA
    if (rc) return rc;
    return launch_grounding_tail(txt, vis, tmask, vmask, marginal, B, Q, V, d, in_dtype, num_token, w_vis2txt, wsf, p, out_sums,
                                 g_txt, g_vis, s);
}

size_t vlg_align_reduced_workspace(int B, int Q) {
    if (B < 1 || Q < 1) return 0;
    return vlg::ReducedPlan(B, Q).bytes;
}

static int reduced_check(const char* what, const void* txt, const void* vis, const float* marginal, int B, int Q, int V, int d,
                         int in_dtype, const void* ws, size_t ws_bytes, size_t need) {
    using namespace vlg;
    if (B < 1 || Q < 1 || V < 1 || d < 1) return set_error(VLG_ERR_SHAPE, "%s: bad shape B=%d Q=%d V=%d d=%d", what, B, Q, V, d);
    if (Q > 65535 || V > 32767) return set_error(VLG_ERR_SHAPE, "%s: Q=%d V=%d exceed the 16-bit position range", what, Q, V);
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "%s: B=%d exceeds grid.y", what, B);
    if (d != 32 && d != 64 && d != 128) return set_error(VLG_ERR_SHAPE, "%s: d=%d (supported: 32, 64, 128)", what, d);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "%s: in_dtype %d", what, in_dtype);
    if (!txt || !vis || !marginal) return set_error(VLG_ERR_ARG, "%s: null buffer", what);
    if (!ws || ws_bytes < need) return set_error(VLG_ERR_WORKSPACE, "%s: workspace %zu bytes < %zu", what, ws_bytes, need);
    return 0;
}

int vlg_align_reduced(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marginal, int B,
                      int Q, int V, int d, int in_dtype, float neg_inf, void* ws, size_t ws_bytes, float* out_logit, void* stream) {
    using namespace vlg;
    const ReducedPlan p(B > 0 ? B : 1, Q > 0 ? Q : 1);
    if (int rc = reduced_check("align_reduced", txt, vis, marginal, B, Q, V, d, in_dtype, ws, ws_bytes, p.bytes)) return rc;
    if (!out_logit) return set_error(VLG_ERR_ARG, "align_reduced: null output");
    hipStream_t s = (hipStream_t)stream;
    float* wsf = (float*)ws;
    AlignArgs xa{nullptr, nullptr, 0, reinterpret_cast<uint16_t*>(wsf + p.off_argV), nullptr};
    const bool f32in = in_dtype == VLG_F32;
    int rc = -1;
#define VLG_RA(F32, KCHV)                                                                                                     \
    rc = V > kCTB * 16                                                                                                        \
             ? launch_align_mfma<F32, KCHV, false, true, 3, true>(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, nullptr,           \
                                                                   wsf + p.off_maxV, nullptr, nullptr, s, xa)                      \
             : launch_align_mfma<F32, KCHV, true, true, 3>(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, nullptr, wsf + p.off_maxV, \
                                                           nullptr, nullptr, s, xa)
    if (!f32in && d == 128 && (V <= kAMRows || !VLG_ENV("VLG_ALIGN_ARGMAX_OLD")) && !VLG_ENV("VLG_GROUND_OLD_ALIGN"))   // row maxima + positions only
        rc = launch_align_max<true>(txt, vis, tmask, vmask, B, B, Q, V, neg_inf, wsf + p.off_maxV, nullptr, s, xa);
    else if (!f32in && d == 128) VLG_RA(false, 4);
    else if (!f32in && d == 64) VLG_RA(false, 2);
    else if (!f32in && d == 32) VLG_RA(false, 1);
    else if (f32in && d == 128) VLG_RA(true, 8);
    else if (f32in && d == 64) VLG_RA(true, 4);
    else VLG_RA(true, 2);
#undef VLG_RA
    if (rc) return rc;
    return launch_reduced_logit(wsf + p.off_maxV, marginal, B, Q, wsf + p.off_sum, out_logit, s);
}

int vlg_align_reduced_backward(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marginal,
                               const float* g_logit, int B, int Q, int V, int d, int in_dtype, void* ws, size_t ws_bytes,
                               float* g_txt, float* g_vis, void* stream) {
    using namespace vlg;
    const ReducedPlan p(B > 0 ? B : 1, Q > 0 ? Q : 1);
    if (int rc = reduced_check("align_reduced_backward", txt, vis, marginal, B, Q, V, d, in_dtype, ws, ws_bytes, p.bytes)) return rc;
    if (!g_logit || (!g_txt && !g_vis)) return set_error(VLG_ERR_ARG, "align_reduced_backward: null buffer");
    return launch_reduced_backward(txt, vis, tmask, vmask, marginal, g_logit, B, Q, V, d, in_dtype, (float*)ws, p, g_txt, g_vis,
                                   (hipStream_t)stream);
}

}  // extern "C"
