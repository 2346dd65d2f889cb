// vlg_align.hip -- region x word alignment kernels (gfx950) and their C-ABI entry points.
//
//   vlg_bilinear_align : DependencyBoxRel.gather_logit_simple  (src/model/joint.py:406-419)
//   vlg_attn_fuse      : the attention-fuse feeding the parser  (src/model/joint.py:670-674)
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

#include "vlg_common.h"
#include "vlg_dp_core.h"   // F32In / BF16In element loaders

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
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kCTB = 3;      // col tiles (16 regions each) per LDS tile pass: 48 regions
constexpr int kTileVP = kCTB * 16 + 4;   // tile pitch 52: rows stay 16-byte aligned, and 52q + 9p hits 32 distinct banks

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

// One block = caption b x (4 waves x a_per_wave images); each wave owns its images and its LDS tile.
// d == KCH * KW exactly (dispatch guarantees it), so fragment loads need no K guards: lane l of a 16-row
// operand tile loads elements [row l&15][kc*KW + EPL*(l>>4) .. +EPL) -- one 16-byte load, and the kc offsets
// are instruction immediates.  Rows past the end are clamped; their products are never stored.
template <bool F32IN, int KCH, bool TILE>
__global__ __launch_bounds__(kAlignThreads) void align_mfma_kernel(
    const typename MfmaCfg<F32IN>::T* __restrict__ txt, const typename MfmaCfg<F32IN>::T* __restrict__ vis,
    const uint8_t* __restrict__ tmask, const uint8_t* __restrict__ vmask, int B, int A, int Q, int V,
    float neg_inf, float* __restrict__ out_full, float* __restrict__ out_maxV, float* __restrict__ out_maxQ,
    float* __restrict__ out_diag, int a_per_wave) {
    using C = MfmaCfg<F32IN>;
    using Frag = typename C::Frag;
    constexpr int RTB = C::RTB, QB = RTB * 16, VB = kCTB * 16, d = KCH * C::KW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b = blockIdx.y;
    float* tile = reinterpret_cast<float*>(smem_raw) + (size_t)wave * (QB * kTileVP + QB);   // [QB][kTileVP]
    float* mv = tile + QB * kTileVP;                                                         // [QB] running max over V
    const int a_begin = (blockIdx.x * 4 + wave) * a_per_wave;
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
        const int n_img = max(0, min(a_per_wave, A - a_begin));
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
        if (n_items > 0) { load_tile(0, 0, f0, k0); load_tile(0, 1, f1, k1); }

        for (int item = 0; item < n_items; ++item) {
            const int ai = item / n_grp, v0 = (item - ai * n_grp) * VB;
            const int a = a_begin + ai, vn = min(VB, V - v0);
            const size_t ob = ((size_t)b * A + a) * Q;   // row base of this (b, a) pair
            float cmax[kCTB];   // running max over this lane's rows, per region column (for max over Q)

            auto compute = [&](int ct, const Frag* bf, unsigned keepv) {
                // rows past Q and masked rows / columns all take the fill value, so one select + one max per element
                const unsigned lane_keep = (v0 + ct * 16 + ccol < V && keepv != 0) ? tkeep : 0u;
                float cm = neg_infinity();
                float* tcol = tile + crow * kTileVP + ct * 16 + ccol;   // + (16*rt + e) * kTileVP: immediate offsets
#pragma unroll
                for (int rt = 0; rt < RTB; ++rt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) acc = mma_chunk<F32IN>(afrag[rt][kc], bf[kc], acc);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float val = (lane_keep >> (rt * 4 + e)) & 1u ? acc[e] : neg_inf;   // joint.py:417-418
                        if (TILE) tcol[(rt * 16 + e) * kTileVP] = val;
                        cm = fmaxf(cm, val);
                    }
                }
                return cm;
            };
            load_tile(item, 2, f2, k2);
            cmax[0] = compute(0, f0, k0);
            load_tile(item + 1, 0, f0, k0);
            cmax[1] = compute(1, f1, k1);
            load_tile(item + 1, 1, f1, k1);
            cmax[2] = compute(2, f2, k2);
            static_assert(kCTB == 3, "the fragment ring is written for three region tiles per group");

            // ---- max over Q straight from the accumulators: lanes l, l^16, l^32, l^48 hold the same column ----
            if (out_maxQ) {
#pragma unroll
                for (int ct = 0; ct < kCTB; ++ct) {
                    float m = cmax[ct];
                    m = fmaxf(m, __shfl_xor(m, 16, 64));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    const int vcol = v0 + ct * 16 + ccol;
                    if (lane < 16 && vcol < V) {
                        float* dst = out_maxQ + ((size_t)b * A + a) * V + vcol;
                        *dst = q0 == 0 ? m : fmaxf(*dst, m);   // same wave handles every row group of (b, a)
                    }
                }
            }
            if (TILE) {
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
                        m = fmaxf(m, __shfl_xor(m, 1, 64));
                        m = fmaxf(m, __shfl_xor(m, 2, 64));
                        if (part == 0 && q < qn) {
                            if (v0 != 0) m = fmaxf(m, mv[q]);
                            if (v0 + VB >= V) out_maxV[ob + q0 + q] = m;
                            else mv[q] = m;
                        }
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
}

template <bool F32IN, int KCH, bool TILE>
static int launch_align_mfma(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A,
                             int Q, int V, float neg_inf, float* out_full, float* out_maxV, float* out_maxQ,
                             float* out_diag, hipStream_t s) {
    using C = MfmaCfg<F32IN>;
    const int a_per_wave = A >= 2048 ? 16 : A >= 64 ? 8 : 1;
    dim3 grid((A + 4 * a_per_wave - 1) / (4 * a_per_wave), B);
    constexpr int QB = C::RTB * 16;
    const size_t lds = TILE ? sizeof(float) * 4 * (size_t)(QB * kTileVP + QB) : 0;
    auto k = align_mfma_kernel<F32IN, KCH, TILE>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, grid, dim3(kAlignThreads), lds, s, (const typename C::T*)txt, (const typename C::T*)vis, tmask,
                       vmask, B, A, Q, V, neg_inf, out_full, out_maxV, out_maxQ, out_diag, a_per_wave);
    return check_launch("align_mfma_kernel");
}

// One block = one sentence b and a chunk of QC words.
//   s[q][v] = vis[b,v,:] . txt[b,1+q,:]  ->  softmax over v (NO region masking: faithful to joint.py:670-672)
//   y[q][c] = enc_x[b,q,c] + sum_v att[q][v] * vis_mid[b,v,c]  ->  LayerNorm over c (biased variance)
constexpr int kFT = 32;   // region rows staged per tile

template <typename In>
__global__ __launch_bounds__(kAlignThreads) void attn_fuse_kernel(
    const typename In::T* __restrict__ vis, const typename In::T* __restrict__ txt,
    const typename In::T* __restrict__ vis_mid, const typename In::T* __restrict__ enc_x,
    const float* __restrict__ gamma, const float* __restrict__ beta, int Lq, int V, int d, int h, float eps, int QC,
    float* __restrict__ out_att, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, b = blockIdx.y;
    const int q0 = blockIdx.x * QC, qn = min(QC, Lq - q0);
    const int ds = d + 1;
    float* txt_s = smem;                 // [QC][ds]
    float* att_s = txt_s + QC * ds;      // [QC][V]   scores -> probabilities
    float* y_s = att_s + QC * V;         // [QC][h]
    float* st_s = y_s + QC * h;          // [QC][2]   mean, rstd
    float* tile = st_s + QC * 2;         // [kFT][max(ds, h)]

    for (int i = tid; i < qn * d; i += kAlignThreads) {
        const int q = i / d, k = i - q * d;
        txt_s[q * ds + k] = In::ld(txt, ((size_t)b * (Lq + 1) + 1 + q0 + q) * d + k);   // skip the root slot, [:, 1:]
    }
    for (int i = tid; i < qn * h; i += kAlignThreads) y_s[i] = In::ld(enc_x, ((size_t)b * Lq + q0) * h + i);
    for (int v0 = 0; v0 < V; v0 += kFT) {
        const int vn = min(kFT, V - v0);
        __syncthreads();
        for (int i = tid; i < vn * d; i += kAlignThreads) {
            const int v = i / d, k = i - v * d;
            tile[v * ds + k] = In::ld(vis, ((size_t)b * V + v0 + v) * d + k);
        }
        __syncthreads();
        for (int i = tid; i < qn * vn; i += kAlignThreads) {
            const int q = i / vn, v = i - q * vn;
            const float* x = txt_s + q * ds;
            const float* y = tile + v * ds;
            float acc = 0.f;
            for (int k = 0; k < d; ++k) acc = fmaf(x[k], y[k], acc);
            att_s[q * V + v0 + v] = acc;
        }
    }
    __syncthreads();
    for (int q = tid; q < qn; q += kAlignThreads) {   // row softmax, one thread per word
        float* s = att_s + q * V;
        float m = s[0];
        for (int v = 1; v < V; ++v) m = fmaxf(m, s[v]);
        float z = 0.f;
        for (int v = 0; v < V; ++v) { const float e = __expf(s[v] - m); s[v] = e; z += e; }
        const float inv = 1.f / z;
        for (int v = 0; v < V; ++v) s[v] *= inv;
    }
    __syncthreads();
    if (out_att)
        for (int i = tid; i < qn * V; i += kAlignThreads) out_att[((size_t)b * Lq + q0) * V + i] = att_s[i];
    // y += att . vis_mid, region tile by region tile; element (q, c) is owned by one thread throughout
    for (int v0 = 0; v0 < V; v0 += kFT) {
        const int vn = min(kFT, V - v0);
        __syncthreads();
        for (int i = tid; i < vn * h; i += kAlignThreads) tile[i] = In::ld(vis_mid, ((size_t)b * V + v0) * h + i);
        __syncthreads();
        for (int i = tid; i < qn * h; i += kAlignThreads) {
            const int q = i / h, c = i - q * h;
            const float* p = att_s + q * V + v0;
            float acc = y_s[i];
            for (int v = 0; v < vn; ++v) acc = fmaf(p[v], tile[v * h + c], acc);
            y_s[i] = acc;
        }
    }
    __syncthreads();
    for (int q = tid; q < qn; q += kAlignThreads) {   // LayerNorm statistics (nn.LayerNorm: biased variance)
        const float* y = y_s + q * h;
        float mean = 0.f;
        for (int c = 0; c < h; ++c) mean += y[(c + q) % h];   // rotate the start: lanes hit different banks
        mean /= (float)h;
        float var = 0.f;
        for (int c = 0; c < h; ++c) { const float t = y[(c + q) % h] - mean; var = fmaf(t, t, var); }
        st_s[q * 2] = mean;
        st_s[q * 2 + 1] = rsqrtf(var / (float)h + eps);
    }
    __syncthreads();
    for (int i = tid; i < qn * h; i += kAlignThreads) {
        const int q = i / h, c = i - q * h;
        out[((size_t)b * Lq + q0) * h + i] = (y_s[i] - st_s[q * 2]) * st_s[q * 2 + 1] * gamma[c] + beta[c];
    }
}

// ---- attention-fuse, fast path -------------------------------------------------------------------------
// Same data flow as attn_fuse_kernel (which stays as the any-shape fallback and for out_att), written for
// throughput: 16-byte LDS reads, register blocking (5 words x 1 region per thread for the scores; one output
// channel x all words per thread for att . vis_mid with the region tile of that channel held in registers),
// every loop fully unrolled so the LDS latency is paid once per tile, not once per element.
// ---- matrix-core path ----------------------------------------------------------------------------------
// One WAVE = one sentence x 16 words; no LDS, no barriers, every wave independent (B * ceil(L/16) waves).
// Both contractions run as v_mfma_f32_16x16x4_f32 (exact fp32 products; bf16 inputs are widened on load), chained
// without moving data between lanes:
//   GEMM 1 computes the TRANSPOSED score tile  S^T[region][word] = vis . txt^T  (A = region rows, B = word rows),
//          whose accumulator layout  lane (r, g), register n  <->  (word r, region 16t + 4g + n)  ...
//   GEMM 2 ... is exactly the B-operand layout of  Y^T[channel][word] = mid^T . P^T  when the K index of MFMA (t, n)
//          is read as region 16t + 4g + n, so the softmaxed accumulators feed the second MFMA directly.
// The softmax over regions and the LayerNorm over channels are both "registers x the four 16-lane groups" reductions
// (two xor-shuffles).  Operand fragments come straight from global memory: a lane's four K values of a chunk are 16
// contiguous bytes, and the dot product does not care that the K order is permuted identically on both operands.
constexpr int kAttnMaxCT = 16;  // channel tiles of 16: h <= 256
constexpr int kAttnKJ = 8;      // 16-feature groups per K chunk (128 features)
constexpr int kAttnPF = 4;      // channel tiles of vis_mid operands in flight ahead of the MFMAs

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const uint16_t* p) {   // four bf16 -> fp32
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
// Buffer-addressed element read: a per-lane offset (VGPR) reused by a run of reads + a uniform offset (SGPR) per read.
__device__ __forceinline__ float buf_ld(F32In, __amdgpu_buffer_rsrc_t r, int lane_off, int uni_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_off * 4, uni_off * 4, 0));
}
__device__ __forceinline__ float buf_ld(BF16In, __amdgpu_buffer_rsrc_t r, int lane_off, int uni_off) {
    const unsigned short u = (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, lane_off * 2, uni_off * 2, 0);
    return __uint_as_float((uint32_t)u << 16);
}
__device__ __forceinline__ float group_sum4(float x) {   // sum over the four 16-lane groups
    x += __shfl_xor(x, 16, 64);
    return x + __shfl_xor(x, 32, 64);
}
__device__ __forceinline__ float group_max4(float x) {
    x = fmaxf(x, __shfl_xor(x, 16, 64));
    return fmaxf(x, __shfl_xor(x, 32, 64));
}

// T (region tiles) is a template parameter and everything else is predicated by clamping, not branching: a uniform
// branch would end the basic block and make every group of operand loads wait out its full latency before the next
// is issued (measured: 31 us with branches).  Channel tiles past h/16 recompute tile h/16-1 and are never stored.
template <typename In, int T>
__global__ __launch_bounds__(64) void attn_fuse_mfma_kernel(
    const typename In::T* __restrict__ vis, const typename In::T* __restrict__ txt,
    const typename In::T* __restrict__ vis_mid, const typename In::T* __restrict__ enc_x,
    const float* __restrict__ gamma, const float* __restrict__ beta, int Lq, int V, int d, int h, float eps,
    float* __restrict__ out) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 16;
    const int CT = h >> 4;
    const int qw = min(q0 + r, Lq - 1);   // this lane's word (clamped; rows past Lq are never stored)
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);   // [16 words][hp]
    const int hp = h + 4;
    float4 erows[16];   // residual rows, needed only after both GEMMs: the read latency is free
#pragma unroll
    for (int i = 0; i < 16; ++i)
        erows[i] = ld4(enc_x + ((size_t)b * Lq + min(q0 + i, Lq - 1)) * h + min(4 * lane, h - 4));

    const typename In::T* trow = txt + ((size_t)b * (Lq + 1) + 1 + qw) * d + 4 * g;   // root slot skipped: txt[:, 1:]
    const __amdgpu_buffer_rsrc_t mid_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(vis_mid + (size_t)b * V * h), 0, (int)(V * h * sizeof(typename In::T)), 0x00020000);
    f32x4 Y[kAttnMaxCT];
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = neg_infinity(), z_run = 0.f;   // running softmax statistics of this lane's word

    // Regions stream through in chunks of 16 T (one chunk when V <= 64, the benchmark's case); between chunks the
    // accumulators are rescaled by exp(old max - new max), the usual streaming softmax.
    for (int v0 = 0; v0 < V; v0 += 16 * T) {
        // ---- GEMM 1: S^T[region v0+16t+4g+n][word r] (joint.py:670-672) ----
        f32x4 S[T];
#pragma unroll
        for (int t = 0; t < T; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const typename In::T* vrow[T];
#pragma unroll
        for (int t = 0; t < T; ++t) vrow[t] = vis + ((size_t)b * V + min(v0 + 16 * t + r, V - 1)) * d + 4 * g;
        for (int k0 = 0; k0 < d; k0 += 16 * kAttnKJ) {
            const int nj = min(kAttnKJ, (d - k0) >> 4);
            float4 wf[kAttnKJ];
#pragma unroll
            for (int j = 0; j < kAttnKJ; ++j) {
                const float4 x = ld4(trow + k0 + 16 * min(j, nj - 1));
                wf[j] = j < nj ? x : zero4;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float4 rf[kAttnKJ];
#pragma unroll
                for (int j = 0; j < kAttnKJ; ++j) {
                    const float4 x = ld4(vrow[t] + k0 + 16 * min(j, nj - 1));
                    rf[j] = j < nj ? x : zero4;
                }
#pragma unroll
                for (int j = 0; j < kAttnKJ; ++j) {
                    S[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[j].x, wf[j].x, S[t], 0, 0, 0);
                    S[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[j].y, wf[j].y, S[t], 0, 0, 0);
                    S[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[j].z, wf[j].z, S[t], 0, 0, 0);
                    S[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(rf[j].w, wf[j].w, S[t], 0, 0, 0);
                }
            }
        }
        // ---- softmax over regions (NO region masking: faithful to joint.py:670-672; only the tile padding is dropped) ----
        float m = m_run;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (v0 + 16 * t + 4 * g + n >= V) S[t][n] = neg_infinity();
                m = fmaxf(m, S[t][n]);
            }
        m = group_max4(m);   // finite: every chunk holds at least one real region
        const float rescale = __expf(m_run - m);   // 0 on the first chunk
        float z = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                S[t][n] = __expf(S[t][n] - m);
                z += S[t][n];
            }
        z_run = z_run * rescale + group_sum4(z);
        m_run = m;
        if (v0 > 0) {
#pragma unroll
            for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= rescale;
        }

        // ---- GEMM 2: Y^T[channel 16ct+4g+n][word r] += mid[region][channel] * exp(score - max) ----
        int mid_lane[T][4];   // region (clamped: padding rows carry zero weight), channel r (+ 16ct uniform)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) mid_lane[t][n] = min(v0 + 16 * t + 4 * g + n, V - 1) * h + r;
        // Operand ring, kAttnPF channel tiles ahead: with ~3 waves per CU nothing else hides the read latency.
        float mv[kAttnPF + 1][T][4];
#pragma unroll
        for (int p = 0; p < kAttnPF; ++p)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n) mv[p][t][n] = buf_ld(In{}, mid_b, mid_lane[t][n], 16 * min(p, CT - 1));
        __builtin_amdgcn_sched_barrier(0);   // keep the ring's issue order: the scheduler otherwise sinks the reads
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) {
            if (ct + kAttnPF < kAttnMaxCT) {   // compile-time
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        mv[(ct + kAttnPF) % (kAttnPF + 1)][t][n] =
                            buf_ld(In{}, mid_b, mid_lane[t][n], 16 * min(ct + kAttnPF, CT - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    Y[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(mv[ct % (kAttnPF + 1)][t][n], S[t][n], Y[ct], 0, 0, 0);
        }
    }
    const float zinv = 1.f / z_run;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= zinv;

    // ---- residual + LayerNorm over channels (biased variance like nn.LayerNorm) ----
    // enc_x rows were requested at kernel start as whole rows (one instruction = one contiguous row); they meet the
    // accumulators in LDS, and the result leaves as whole rows again.  Reading / writing in accumulator layout directly
    // moves 64-byte pieces.
#pragma unroll
    for (int i = 0; i < 16; ++i)
        if (4 * lane < h) *reinterpret_cast<float4*>(tile + i * hp + 4 * lane) = erows[i];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) {
        float* cell = tile + r * hp + 16 * min(ct, CT - 1) + 4 * g;
        const float4 e = *reinterpret_cast<const float4*>(cell);
        const float keep = ct < CT ? 1.f : 0.f;
        Y[ct][0] += e.x; Y[ct][1] += e.y; Y[ct][2] += e.z; Y[ct][3] += e.w;
#pragma unroll
        for (int n = 0; n < 4; ++n) { s1 = fmaf(keep, Y[ct][n], s1); s2 = fmaf(keep * Y[ct][n], Y[ct][n], s2); }
        if (ct < CT) *reinterpret_cast<float4*>(cell) = make_float4(Y[ct][0], Y[ct][1], Y[ct][2], Y[ct][3]);
    }
    s1 = group_sum4(s1);
    s2 = group_sum4(s2);
    const float mean = s1 / (float)h;
    const float rstd = rsqrtf(fmaxf(s2 / (float)h - mean * mean, 0.f) + eps);
    const int cl = min(4 * lane, h - 4);
    const float4 gm = *reinterpret_cast<const float4*>(gamma + cl);
    const float4 bt = *reinterpret_cast<const float4*>(beta + cl);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float mu = __shfl(mean, i, 64), rs = __shfl(rstd, i, 64);   // word i's statistics live in lanes r == i
        const float4 y = *reinterpret_cast<const float4*>(tile + i * hp + cl);
        float4 o;
        o.x = (y.x - mu) * rs * gm.x + bt.x;
        o.y = (y.y - mu) * rs * gm.y + bt.y;
        o.z = (y.z - mu) * rs * gm.z + bt.z;
        o.w = (y.w - mu) * rs * gm.w + bt.w;
        if (q0 + i < Lq && 4 * lane < h) *reinterpret_cast<float4*>(out + ((size_t)b * Lq + q0 + i) * h + cl) = o;
    }
}

template <typename In, int T>
static void launch_attn_mfma(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                             const float* beta, int B, int L, int V, int d, int h, float eps, float* out, hipStream_t s) {
    using P = const typename In::T*;
    hipLaunchKernelGGL((attn_fuse_mfma_kernel<In, T>), dim3((L + 15) / 16, B), dim3(64), sizeof(float) * 16 * (h + 4), s, (P)vis, (P)txt, (P)vis_mid,
                       (P)enc_x, gamma, beta, L, V, d, h, eps, out);
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
#define VLG_MFMA(F32, KCHV)                                                                                         \
    return tile ? launch_align_mfma<F32, KCHV, true>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, out_full, out_maxV, \
                                                     out_maxQ, out_diag, s)                                           \
                : launch_align_mfma<F32, KCHV, false>(txt, vis, tmask, vmask, B, A, Q, V, neg_inf, out_full, out_maxV, \
                                                      out_maxQ, out_diag, s)
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

int vlg_attn_fuse(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                  const float* beta, int B, int L, int V, int d, int h, int in_dtype, float eps, float* out_att,
                  float* out, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || V < 1 || d < 1 || h < 1)
        return set_error(VLG_ERR_SHAPE, "attn_fuse: bad shape B=%d L=%d V=%d d=%d h=%d", B, L, V, d, h);
    if (B == 0) return 0;
    if (!vis || !txt || !vis_mid || !enc_x || !gamma || !beta || !out) return set_error(VLG_ERR_ARG, "attn_fuse: null buffer");
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "attn_fuse: B=%d exceeds grid.y", B);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "attn_fuse: in_dtype %d", in_dtype);
    // ---- matrix-core path: one wave per 16 words, no attention map requested ----
    if (!out_att && d % 16 == 0 && h % 16 == 0 && h <= 16 * kAttnMaxCT && (size_t)V * h * 4 < (1u << 31)) {
#define VLG_ATTN(INV)                                                                                              \
    switch (V > 48 ? 4 : (V + 15) / 16) { /* region tiles per chunk; V > 64 streams chunks of 64 */                  \
        case 1: launch_attn_mfma<INV, 1>(vis, txt, vis_mid, enc_x, gamma, beta, B, L, V, d, h, eps, out, s); break; \
        case 2: launch_attn_mfma<INV, 2>(vis, txt, vis_mid, enc_x, gamma, beta, B, L, V, d, h, eps, out, s); break; \
        case 3: launch_attn_mfma<INV, 3>(vis, txt, vis_mid, enc_x, gamma, beta, B, L, V, d, h, eps, out, s); break; \
        default: launch_attn_mfma<INV, 4>(vis, txt, vis_mid, enc_x, gamma, beta, B, L, V, d, h, eps, out, s); break; \
    }
        if (in_dtype == VLG_F32) { VLG_ATTN(F32In) } else { VLG_ATTN(BF16In) }
#undef VLG_ATTN
        return check_launch("attn_fuse_mfma_kernel");
    }
    const size_t tile_f = (size_t)kFT * (size_t)((d + 1) > h ? (d + 1) : h);
    const size_t per_q = (size_t)(d + 1) + V + h + 2;
    int QC = L < 32 ? L : 32;
    while (QC > 1 && sizeof(float) * (tile_f + per_q * QC) > 150 * 1024) QC >>= 1;
    const size_t lds = sizeof(float) * (tile_f + per_q * QC);
    if (lds > 150 * 1024) return set_error(VLG_ERR_SHAPE, "attn_fuse: V=%d d=%d h=%d exceed the LDS budget", V, d, h);
#define VLG_LAUNCH(INV)                                                                                            \
    do {                                                                                                           \
        auto k = attn_fuse_kernel<INV>;                                                                            \
        if (lds > 60 * 1024) {                                                                                     \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);             \
            if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));        \
        }                                                                                                          \
        hipLaunchKernelGGL(k, dim3((L + QC - 1) / QC, B), dim3(kAlignThreads), lds, s, (const INV::T*)vis,          \
                           (const INV::T*)txt, (const INV::T*)vis_mid, (const INV::T*)enc_x, gamma, beta, L, V, d, \
                           h, eps, QC, out_att, out);                                                              \
    } while (0)
    if (in_dtype == VLG_F32) VLG_LAUNCH(F32In);
    else VLG_LAUNCH(BF16In);
#undef VLG_LAUNCH
    return check_launch("attn_fuse_kernel");
}

}  // extern "C"
