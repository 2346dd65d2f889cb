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
constexpr int kQCap = 40;   // words per block (register-resident accumulators)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const uint16_t* p) {   // four bf16 -> fp32
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <typename In>
__global__ __launch_bounds__(kAlignThreads) void attn_fuse_fast_kernel(
    const typename In::T* __restrict__ vis, const typename In::T* __restrict__ txt,
    const typename In::T* __restrict__ vis_mid, const typename In::T* __restrict__ enc_x,
    const float* __restrict__ gamma, const float* __restrict__ beta, int Lq, int V, int d, int h, float eps,
    float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y;
    const int q0 = blockIdx.x * kQCap, qn = min(kQCap, Lq - q0);
    const int dp = d + 4, hp = h + 4, Vp = ((V + 31) & ~31) + 4;   // rows padded to whole 32-region tiles
    // LDS: [att_s QCap x Vp][st_s QCap x 2][region (union): txt_s QCap x dp + vis tile 32 x dp | mid tile 32 x hp | y_s QCap x hp]
    float* att_s = smem;
    float* st_s = att_s + kQCap * Vp;
    float* region = st_s + kQCap * 2;
    float* txt_s = region;
    float* vtile = region + kQCap * dp;

    // ---- words of this block (root slot skipped: txt[:, 1:], joint.py:671) ----
    for (int i = tid; i < qn * (d >> 2); i += kAlignThreads) {
        const int q = i / (d >> 2), k4 = i - q * (d >> 2);
        *reinterpret_cast<float4*>(txt_s + q * dp + k4 * 4) = ld4(txt + ((size_t)b * (Lq + 1) + 1 + q0 + q) * d + k4 * 4);
    }
    // ---- scores: thread = (region v of the tile, word group qg); words qg, qg+8, ... ----
    const int sv = tid & 31, qg = tid >> 5;
    for (int v0 = 0; v0 < V; v0 += 32) {
        const int vn = min(32, V - v0);
        __syncthreads();
        for (int i = tid; i < vn * (d >> 2); i += kAlignThreads) {
            const int v = i / (d >> 2), k4 = i - v * (d >> 2);
            *reinterpret_cast<float4*>(vtile + v * dp + k4 * 4) = ld4(vis + ((size_t)b * V + v0 + v) * d + k4 * 4);
        }
        __syncthreads();
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const float* yrow = vtile + min(sv, vn - 1) * dp;
        for (int k4 = 0; k4 < (d >> 2); ++k4) {
            const float4 y = *reinterpret_cast<const float4*>(yrow + k4 * 4);
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const float4 x = *reinterpret_cast<const float4*>(txt_s + min(qg + 8 * u, qn - 1) * dp + k4 * 4);
                acc[u] = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, acc[u]))));
            }
        }
        if (sv < vn)
#pragma unroll
            for (int u = 0; u < 5; ++u)
                if (qg + 8 * u < qn) att_s[(qg + 8 * u) * Vp + v0 + sv] = acc[u];
    }
    __syncthreads();
    // ---- softmax over regions: 4 lanes per word (NO region masking: faithful to joint.py:670-672) ----
    {
        const int q = tid >> 2, part = tid & 3, vq = (V + 3) >> 2, lo = part * vq, hi = min(V, lo + vq);
        if (q < kQCap) {   // 160 lanes, whole waves only diverge at the tail
            float* row = att_s + min(q, qn - 1) * Vp;
            float m = neg_infinity();
            for (int v = lo; v < hi; ++v) m = fmaxf(m, row[v]);
            m = fmaxf(m, __shfl_xor(m, 1, 64));
            m = fmaxf(m, __shfl_xor(m, 2, 64));
            float z = 0.f;
            for (int v = lo; v < hi; ++v) z += __expf(row[v] - m);
            z += __shfl_xor(z, 1, 64);
            z += __shfl_xor(z, 2, 64);
            const float inv = 1.f / z;
            if (q < qn)
                for (int v = lo; v < hi; ++v) row[v] = __expf(row[v] - m) * inv;
        }
    }
    // zero the padding columns so that whole 32-region tiles can be consumed below
    for (int i = tid; i < qn * (Vp - V); i += kAlignThreads) {
        const int q = i / (Vp - V), v = V + (i - q * (Vp - V));
        att_s[q * Vp + v] = 0.f;
    }
    // ---- y = att . vis_mid : thread owns output channels c = tid (+256, ...), all words in registers ----
    float* mtile = region;   // [32][hp]   (txt_s / vis tile are dead from here on)
    for (int c0 = 0; c0 < h; c0 += kAlignThreads) {
        const int c = c0 + tid;
        const bool cin = c < h;
        float yacc[kQCap];
#pragma unroll
        for (int q = 0; q < kQCap; ++q) yacc[q] = 0.f;
        for (int v0 = 0; v0 < V; v0 += 32) {
            const int vn = min(32, V - v0);
            __syncthreads();
            for (int i = tid; i < 32 * (h >> 2); i += kAlignThreads) {   // rows past vn are zero-filled
                const int v = i / (h >> 2), c4 = i - v * (h >> 2);
                const float4 val = v < vn ? ld4(vis_mid + ((size_t)b * V + v0 + v) * h + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(mtile + v * hp + c4 * 4) = val;
            }
            __syncthreads();
            float mreg[32];
#pragma unroll
            for (int v = 0; v < 32; ++v) mreg[v] = mtile[v * hp + (cin ? c : 0)];
#pragma unroll
            for (int q = 0; q < kQCap; ++q) {
                if (q < qn) {   // block-uniform
                    const float* arow = att_s + q * Vp + v0;
                    float a = yacc[q];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float4 p = *reinterpret_cast<const float4*>(arow + 4 * j);   // broadcast read
                        a = fmaf(p.x, mreg[4 * j], fmaf(p.y, mreg[4 * j + 1], fmaf(p.z, mreg[4 * j + 2], fmaf(p.w, mreg[4 * j + 3], a))));
                    }
                    yacc[q] = a;
                }
            }
        }
        // residual: y = enc_x + att . vis_mid, parked in LDS for the row statistics
        __syncthreads();
        float* y_s = region;   // [QCap][hp]
#pragma unroll
        for (int q = 0; q < kQCap; ++q)
            if (q < qn && cin) y_s[q * hp + c] = yacc[q] + In::ld(enc_x, ((size_t)b * Lq + q0 + q) * h + c);
        // (for h > 256 the loop over c0 would overwrite y_s before the statistics: handled by the launcher, h <= 256)
    }
    __syncthreads();
    // ---- LayerNorm statistics: wave w takes words w, w+4, ...; biased variance like nn.LayerNorm ----
    {
        const float* y_s = region;
        for (int q = wave; q < qn; q += 4) {
            float s1 = 0.f, s2 = 0.f;
            for (int c = lane; c < h; c += 64) { const float t = y_s[q * hp + c]; s1 += t; s2 = fmaf(t, t, s2); }
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) { s1 += __shfl_xor(s1, k, 64); s2 += __shfl_xor(s2, k, 64); }
            if (lane == 0) {
                const float mean = s1 / (float)h;
                st_s[q * 2] = mean;
                st_s[q * 2 + 1] = rsqrtf(fmaxf(s2 / (float)h - mean * mean, 0.f) + eps);
            }
        }
    }
    __syncthreads();
    {
        const float* y_s = region;
        for (int i = tid; i < qn * h; i += kAlignThreads) {
            const int q = i / h, c = i - q * h;
            out[((size_t)b * Lq + q0) * h + i] = (y_s[q * hp + c] - st_s[q * 2]) * st_s[q * 2 + 1] * gamma[c] + beta[c];
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
    // ---- fast path: 16-byte aligned rows, one pass over the output channels, no attention map requested ----
    if (!out_att && d % 4 == 0 && h % 4 == 0 && h <= kAlignThreads) {
        const size_t dp = d + 4, hp = h + 4, Vp = ((V + 31) & ~31) + 4;
        size_t region = (size_t)kQCap * dp + 32 * dp;
        if (32 * hp > region) region = 32 * hp;
        if ((size_t)kQCap * hp > region) region = (size_t)kQCap * hp;
        const size_t lds_fast = sizeof(float) * ((size_t)kQCap * Vp + kQCap * 2 + region);
        if (lds_fast <= 156 * 1024) {
            dim3 grid((L + kQCap - 1) / kQCap, B);
#define VLG_FAST(INV)                                                                                              \
    do {                                                                                                           \
        auto k = attn_fuse_fast_kernel<INV>;                                                                       \
        if (lds_fast > 60 * 1024) {                                                                                \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_fast);        \
            if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));        \
        }                                                                                                          \
        hipLaunchKernelGGL(k, grid, dim3(kAlignThreads), lds_fast, s, (const INV::T*)vis, (const INV::T*)txt,      \
                           (const INV::T*)vis_mid, (const INV::T*)enc_x, gamma, beta, L, V, d, h, eps, out);       \
    } while (0)
            if (in_dtype == VLG_F32) VLG_FAST(F32In);
            else VLG_FAST(BF16In);
#undef VLG_FAST
            return check_launch("attn_fuse_fast_kernel");
        }
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
