// vlg_attn.hip -- the attention-fuse feeding the parser (gfx950) and its C-ABI entry points.
//
//   vlg_attn_fuse           : DependencyBoxRel._forward, src/model/joint.py:670-674
//   vlg_attn_fuse_backward  : its adjoint (what autograd derives for those lines), for training through the fuse
//
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_dp_core.h"   // F32In / BF16In element loaders

namespace vlg {

constexpr int kAlignThreads = 256;   // generic kernel block size
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float neg_infinity() { return __uint_as_float(0xff800000u); }

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

// GEMM 1 of one region chunk: S^T[region v0+16t+4g+n][word r] = <vis[region], txt[word]>  (joint.py:670-672).
// trow = this lane's word row + 4g.
template <typename In, int T>
__device__ __forceinline__ void attn_scores(f32x4 (&S)[T], const typename In::T* trow, const typename In::T* vis_b, int V,
                                            int d, int v0, int r, int g) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < T; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const typename In::T* vrow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) vrow[t] = vis_b + (size_t)min(v0 + 16 * t + r, V - 1) * d + 4 * g;
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
}

// The whole forward of one 16-word tile up to (not including) the residual: on return Y^T[channel 16ct+4g+n][word r]
// holds softmax_v(S) . vis_mid and (m_run, z_run) the softmax statistics of word r.
// Regions stream through in chunks of 16 T (one chunk when V <= 64, the benchmark's case); between chunks the
// accumulators are rescaled by exp(old max - new max), the usual streaming softmax.
template <typename In, int T>
__device__ __forceinline__ void attn_forward_tile(f32x4 (&Y)[kAttnMaxCT], float& m_run, float& z_run,
                                                  const typename In::T* trow, const typename In::T* vis_b,
                                                  const typename In::T* mid_ptr, int V, int d, int h, int r, int g) {
    const int CT = h >> 4;
    const __amdgpu_buffer_rsrc_t mid_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)mid_ptr, 0, (int)(V * h * sizeof(typename In::T)), 0x00020000);
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run = neg_infinity();
    z_run = 0.f;
    for (int v0 = 0; v0 < V; v0 += 16 * T) {
        f32x4 S[T];
        attn_scores<In, T>(S, trow, vis_b, V, d, v0, r, g);
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
}

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
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);   // [16 words][hp]
    const int hp = h + 4;
    float4 erows[16];   // residual rows, needed only after both GEMMs: the read latency is free
#pragma unroll
    for (int i = 0; i < 16; ++i)
        erows[i] = ld4(enc_x + ((size_t)b * Lq + min(q0 + i, Lq - 1)) * h + min(4 * lane, h - 4));

    f32x4 Y[kAttnMaxCT];
    float m_run, z_run;
    attn_forward_tile<In, T>(Y, m_run, z_run, txt + ((size_t)b * (Lq + 1) + 1 + qw) * d + 4 * g,   // root slot skipped: txt[:, 1:]
                             vis + (size_t)b * V * d, vis_mid + (size_t)b * V * h, V, d, h, r, g);

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

// ---- adjoint (training through the fuse) ------------------------------------------------------------------
//   out = yhat * gamma + beta,  yhat = (y - mean) * rstd,  y = enc_x + M,  M = P . mid,  P = softmax_v(S),  S = t . vis^T
//   d_beta = sum dout            d_gamma = sum dout * yhat            dyhat = dout * gamma
//   dy = rstd * (dyhat - mean_c(dyhat) - yhat * mean_c(dyhat * yhat))        d_enc_x = dy
//   dP = dy . mid^T              dS = P o (dP - D),  D = sum_v P o dP = sum_c dy o M       (the flash-attention identity:
//                                                                              no pass over all regions needed for D)
//   d_txt[1+q] = dS . vis        d_vis = dS^T . t            d_mid = P^T . dy
// Two kernels, split by what the sums run over:
//   words kernel   (wave = sentence x 16 words): recompute the forward tile, LayerNorm adjoint, d_enc_x, dP (GEMM 3,
//                  K = channels; its B operand is dy in accumulator layout -- the same register chain as the forward),
//                  dS, d_txt (GEMM 4, K = regions; B operand = dS in accumulator layout), and P^T / dS^T to scratch;
//   regions kernel (wave = sentence x 16 regions): d_mid^T = dy^T . P and d_vis^T = t^T . dS, K = words, operands read
//                  from the scratch / d_enc_x in fragment order.
// d_gamma / d_beta: per-(sentence, word tile) partial rows, then a fixed-order column sum (bit-reproducible; no atomics).
constexpr int kAttnMaxFT = 16;   // feature tiles of 16: d <= 256 in the adjoint

template <typename In, int T, int FTM>
__global__ __launch_bounds__(64) void attn_fuse_bwd_words_kernel(
    const typename In::T* __restrict__ vis, const typename In::T* __restrict__ txt,
    const typename In::T* __restrict__ vis_mid, const typename In::T* __restrict__ enc_x,
    const float* __restrict__ gamma, const float* __restrict__ dout, size_t ld_dout_b, size_t ld_dout_l, int Lq, int V, int d, int h, float eps,
    float* __restrict__ PT, float* __restrict__ DST, int Vp, int Lp, float* __restrict__ part,
    float* __restrict__ d_txt, float* __restrict__ d_enc) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 16;
    const int CT = h >> 4, FT = d >> 4;
    const int qw = min(q0 + r, Lq - 1);
    const bool live = q0 + r < Lq;
    const float livef = live ? 1.f : 0.f;
    const int hp = h + 4, cl = min(4 * lane, h - 4);
    const bool cin = 4 * lane < h;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile_x = reinterpret_cast<float*>(smem_raw);   // [16][hp]: enc_x rows, later dy rows
    float* tile_g = tile_x + 16 * hp;                      // [16][hp]: dout rows, later dout * yhat rows
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    if (blockIdx.x == 0)   // the root slot never enters the scores: its gradient is zero
        for (int k = 4 * lane; k < d; k += 256) *reinterpret_cast<float4*>(d_txt + (size_t)b * (Lq + 1) * d + k) = zero4;

    const typename In::T* trow = txt + ((size_t)b * (Lq + 1) + 1 + qw) * d + 4 * g;
    const typename In::T* vis_b = vis + (size_t)b * V * d;
    const typename In::T* mid_p = vis_mid + (size_t)b * V * h;
    f32x4 Y[kAttnMaxCT];
    float m_run, z_run;
    attn_forward_tile<In, T>(Y, m_run, z_run, trow, vis_b, mid_p, V, d, h, r, g);   // Y = M
    const float zinv = 1.f / z_run;

    // ---- LayerNorm adjoint, in accumulator layout (lane (r,g), register (ct,n) <-> word r, channel 16ct+4g+n) ----
    // enc_x / dout rows are read here, not at kernel start: 128 more live registers across the forward tile spill.
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int qi = min(q0 + i, Lq - 1);
        const size_t row = ((size_t)b * Lq + qi) * h + cl;
        const float4 xr = ld4(enc_x + row);
        const float4 gr = *reinterpret_cast<const float4*>(dout + (size_t)b * ld_dout_b + (size_t)qi * ld_dout_l + cl);
        if (cin) {
            *reinterpret_cast<float4*>(tile_x + i * hp + 4 * lane) = xr;
            *reinterpret_cast<float4*>(tile_g + i * hp + 4 * lane) = gr;
        }
    }
    const size_t prow = ((size_t)b * gridDim.x + blockIdx.x) * 2 * h;   // this tile's partial rows: [d_gamma | d_beta]
    {   // d_beta partial: column sums of dout over the live words of the tile
        float4 acc = zero4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float k = q0 + i < Lq ? 1.f : 0.f;
            const float4 v = *reinterpret_cast<const float4*>(tile_g + i * hp + cl);
            acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
        }
        if (cin) *reinterpret_cast<float4*>(part + prow + h + 4 * lane) = acc;
    }
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) {   // y = M + x; Y keeps y
        const float4 e = *reinterpret_cast<const float4*>(tile_x + r * hp + 16 * min(ct, CT - 1) + 4 * g);
        const float keep = ct < CT ? 1.f : 0.f;
        Y[ct][0] += e.x; Y[ct][1] += e.y; Y[ct][2] += e.z; Y[ct][3] += e.w;
#pragma unroll
        for (int n = 0; n < 4; ++n) { s1 = fmaf(keep, Y[ct][n], s1); s2 = fmaf(keep * Y[ct][n], Y[ct][n], s2); }
    }
    s1 = group_sum4(s1);
    s2 = group_sum4(s2);
    const float mean = s1 / (float)h;
    const float rstd = rsqrtf(fmaxf(s2 / (float)h - mean * mean, 0.f) + eps);
    // D = sum_c dy*M = rstd * (sum dyhat*M - c1 * sum M - c2 * sum yhat*M): all five sums in one sweep.
    f32x4 DY[kAttnMaxCT];
    float c1 = 0.f, c2 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) {
        const int c0 = 16 * min(ct, CT - 1) + 4 * g;
        const float keep = ct < CT ? 1.f : 0.f;
        float* gcell = tile_g + r * hp + c0;
        const float4 e = *reinterpret_cast<const float4*>(tile_x + r * hp + c0);
        const float4 go = *reinterpret_cast<const float4*>(gcell);
        const float4 gm = *reinterpret_cast<const float4*>(gamma + c0);
        const float ev[4] = {e.x, e.y, e.z, e.w}, gov[4] = {go.x, go.y, go.z, go.w}, gmv[4] = {gm.x, gm.y, gm.z, gm.w};
        float gy[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float y = Y[ct][n], M = y - ev[n], yh = (y - mean) * rstd, dyh = gov[n] * gmv[n] * keep;
            c1 += dyh;
            c2 = fmaf(dyh, yh, c2);
            a1 = fmaf(dyh, M, a1);
            a2 = fmaf(keep, M, a2);
            a3 = fmaf(keep * yh, M, a3);
            Y[ct][n] = yh;
            DY[ct][n] = dyh;
            gy[n] = gov[n] * yh;
        }
        if (ct < CT) *reinterpret_cast<float4*>(gcell) = make_float4(gy[0], gy[1], gy[2], gy[3]);
    }
    c1 = group_sum4(c1) / (float)h;
    c2 = group_sum4(c2) / (float)h;
    a1 = group_sum4(a1);
    a2 = group_sum4(a2);
    a3 = group_sum4(a3);
    const float Dsum = rstd * (a1 - c1 * a2 - c2 * a3);
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) {
        const float keep = ct < CT ? 1.f : 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) DY[ct][n] = keep * rstd * (DY[ct][n] - c1 - Y[ct][n] * c2);   // dy
        if (ct < CT)
            *reinterpret_cast<float4*>(tile_x + r * hp + 16 * ct + 4 * g) = make_float4(DY[ct][0], DY[ct][1], DY[ct][2], DY[ct][3]);
    }
    {   // d_gamma partial (column sums of dout*yhat) and d_enc_x rows, both as whole rows
        float4 acc = zero4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool lv = q0 + i < Lq;
            const float k = lv ? 1.f : 0.f;
            const float4 v = *reinterpret_cast<const float4*>(tile_g + i * hp + cl);
            acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
            const float4 dyr = *reinterpret_cast<const float4*>(tile_x + i * hp + cl);
            if (lv && cin) *reinterpret_cast<float4*>(d_enc + ((size_t)b * Lq + q0 + i) * h + 4 * lane) = dyr;
        }
        if (cin) *reinterpret_cast<float4*>(part + prow + 4 * lane) = acc;
    }

    // ---- second sweep over the region chunks: P again, dP, dS, d_txt; P^T and dS^T to scratch ----
    f32x4 dT[FTM];
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft) dT[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t vis_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)vis_b, 0, (int)(V * d * sizeof(typename In::T)), 0x00020000);
    for (int v0 = 0; v0 < V; v0 += 16 * T) {
        f32x4 S[T], dP[T];
        attn_scores<In, T>(S, trow, vis_b, V, d, v0, r, g);
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                S[t][n] = v0 + 16 * t + 4 * g + n < V ? __expf(S[t][n] - m_run) * zinv : 0.f;   // P
            dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // GEMM 3: dP^T[region][word] = sum_c mid[region][c] * dy[word][c]; A fragment = 16 contiguous bytes of a mid row.
        // All region tiles advance together over the channel tiles, their fragments kAttnPF tiles ahead in a register ring.
        const typename In::T* mrow[T];
#pragma unroll
        for (int t = 0; t < T; ++t) mrow[t] = mid_p + (size_t)min(v0 + 16 * t + r, V - 1) * h + 4 * g;
        float4 mf[kAttnPF + 1][T];
#pragma unroll
        for (int pfi = 0; pfi < kAttnPF; ++pfi)
#pragma unroll
            for (int t = 0; t < T; ++t) mf[pfi][t] = ld4(mrow[t] + 16 * min(pfi, CT - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) {
            if (ct + kAttnPF < kAttnMaxCT) {   // compile-time
#pragma unroll
                for (int t = 0; t < T; ++t) mf[(ct + kAttnPF) % (kAttnPF + 1)][t] = ld4(mrow[t] + 16 * min(ct + kAttnPF, CT - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t) {   // dy of tiles past CT is zero
                const float4 a = mf[ct % (kAttnPF + 1)][t];
                dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, DY[ct][0], dP[t], 0, 0, 0);
                dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, DY[ct][1], dP[t], 0, 0, 0);
                dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, DY[ct][2], dP[t], 0, 0, 0);
                dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, DY[ct][3], dP[t], 0, 0, 0);
            }
        }
        int vlane[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int v = v0 + 16 * t + 4 * g + n;
                const float p = S[t][n] * livef;                       // words past Lq contribute nothing downstream
                const float ds = p * (dP[t][n] - Dsum);
                S[t][n] = ds;
                PT[((size_t)b * Vp + v) * Lp + q0 + r] = p;            // v < Vp by construction; zero for v >= V
                DST[((size_t)b * Vp + v) * Lp + q0 + r] = ds;
                vlane[t][n] = min(v, V - 1) * d + r;
            }
        // GEMM 4: d_txt^T[feature][word] += sum_v vis[v][feature] * dS[word][v]; B operand = dS as it sits
        // operand ring kAttnPF feature tiles ahead, as in the forward's second contraction
        float vv[kAttnPF + 1][T][4];
#pragma unroll
        for (int pfi = 0; pfi < kAttnPF; ++pfi)
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n) vv[pfi][t][n] = buf_ld(In{}, vis_rs, vlane[t][n], 16 * min(pfi, FT - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft) {
            if (ft + kAttnPF < FTM) {   // compile-time
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        vv[(ft + kAttnPF) % (kAttnPF + 1)][t][n] = buf_ld(In{}, vis_rs, vlane[t][n], 16 * min(ft + kAttnPF, FT - 1));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    dT[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[ft % (kAttnPF + 1)][t][n], S[t][n], dT[ft], 0, 0, 0);
        }
    }
    if (live) {
        float* orow = d_txt + ((size_t)b * (Lq + 1) + 1 + q0 + r) * d + 4 * g;
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft)
            if (ft < FT) *reinterpret_cast<float4*>(orow + 16 * ft) = make_float4(dT[ft][0], dT[ft][1], dT[ft][2], dT[ft][3]);
    }
}

// wave = sentence x 16 regions.  d_mid^T[channel][region] = sum_w dy[w][channel] * P[w][region]   (A = dy^T, B = P)
//                                d_vis^T[feature][region] = sum_w t[w][feature]  * dS[w][region]  (A = t^T,  B = dS)
// B fragments are 16 contiguous bytes of a P^T / dS^T scratch row (words are the K index); A values are single
// elements of d_enc_x / txt rows, read through buffer descriptors.
template <typename In, int FTM>
__global__ __launch_bounds__(64) void attn_fuse_bwd_regions_kernel(
    const typename In::T* __restrict__ txt, const float* __restrict__ d_enc, const float* __restrict__ PT,
    const float* __restrict__ DST, int Vp, int Lp, int Lq, int V, int d, int h, float* __restrict__ d_mid,
    float* __restrict__ d_vis) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, v = blockIdx.x * 16 + r;   // v < Vp
    const int CT = h >> 4, FT = d >> 4;
    const __amdgpu_buffer_rsrc_t dy_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)(d_enc + (size_t)b * Lq * h), 0, (int)(Lq * h * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t t_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(txt + ((size_t)b * (Lq + 1) + 1) * d), 0, (int)(Lq * d * sizeof(typename In::T)), 0x00020000);
    f32x4 dM[kAttnMaxCT], dV[FTM];
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) dM[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft) dV[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* prow = PT + ((size_t)b * Vp + v) * Lp + 4 * g;
    const float* srow = DST + ((size_t)b * Vp + v) * Lp + 4 * g;
    for (int w0 = 0; w0 < Lp; w0 += 16) {
        const float4 pf = *reinterpret_cast<const float4*>(prow + w0);
        const float4 sf = *reinterpret_cast<const float4*>(srow + w0);
        const float pv[4] = {pf.x, pf.y, pf.z, pf.w}, sv[4] = {sf.x, sf.y, sf.z, sf.w};
        int wl_h[4], wl_d[4];   // word 16j + 4g + e (clamped: the scratch holds zeros past Lq), element r of its row
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int w = min(w0 + 4 * g + e, Lq - 1);
            wl_h[e] = w * h + r;
            wl_d[e] = w * d + r;
        }
        // every A value of this word chunk is requested before the first MFMA (one round trip per chunk, not per tile)
        float dyv[kAttnMaxCT][4], tv[FTM][4];
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) dyv[ct][e] = buf_ld(F32In{}, dy_rs, wl_h[e], 16 * min(ct, CT - 1));
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft)
#pragma unroll
            for (int e = 0; e < 4; ++e) tv[ft][e] = buf_ld(In{}, t_rs, wl_d[e], 16 * min(ft, FT - 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) dM[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(dyv[ct][e], pv[e], dM[ct], 0, 0, 0);
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft)
#pragma unroll
            for (int e = 0; e < 4; ++e) dV[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(tv[ft][e], sv[e], dV[ft], 0, 0, 0);
    }
    if (v < V) {   // accumulator layout: lane (r,g), register n <-> region v (col r), channel / feature 16ct + 4g + n
        float* mrow = d_mid + ((size_t)b * V + v) * h + 4 * g;
        float* vrow = d_vis + ((size_t)b * V + v) * d + 4 * g;
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct)
            if (ct < CT) *reinterpret_cast<float4*>(mrow + 16 * ct) = make_float4(dM[ct][0], dM[ct][1], dM[ct][2], dM[ct][3]);
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft)
            if (ft < FT) *reinterpret_cast<float4*>(vrow + 16 * ft) = make_float4(dV[ft][0], dV[ft][1], dV[ft][2], dV[ft][3]);
    }
}

// Column sums of the partial rows [R][2h] -> d_gamma [h] | d_beta [h], rows added in a fixed order
// (16 interleaved row groups per column, then a fixed tree over the groups).
__global__ __launch_bounds__(1024) void attn_fuse_bwd_affine_kernel(const float* __restrict__ part, int R, int h,
                                                                   float* __restrict__ d_gamma, float* __restrict__ d_beta) {
    __shared__ float red[1024];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < 2 * h) {
#pragma unroll 4
        for (int row = rg; row < R; row += 16) acc += part[(size_t)row * 2 * h + col];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int half = 8; half >= 1; half >>= 1) {
        if (rg < half) red[threadIdx.x] += red[threadIdx.x + 64 * half];
        __syncthreads();
    }
    if (rg == 0 && col < 2 * h) {
        if (col < h) d_gamma[col] = red[threadIdx.x];
        else d_beta[col - h] = red[threadIdx.x];
    }
}

struct AttnBwdPlan {   // scratch carving shared by the size query and the launcher
    int WT, T, Vp, Lp;
    size_t pt_floats, part_floats, bytes;
    AttnBwdPlan(int B, int L, int V, int h) {
        WT = (L + 15) / 16;
        T = V > 48 ? 4 : (V + 15) / 16;
        Vp = ((V + 16 * T - 1) / (16 * T)) * 16 * T;
        Lp = 16 * WT;
        pt_floats = (size_t)B * Vp * Lp;
        part_floats = (size_t)B * WT * 2 * h;
        bytes = sizeof(float) * (2 * pt_floats + part_floats);
    }
};

template <typename In, int T, int FTM>
static void launch_attn_bwd_words(const void* vis, const void* txt, const void* vis_mid, const void* enc_x,
                                  const float* gamma, const float* dout, size_t ld_dout_b, size_t ld_dout_l, int B, int L, int V, int d, int h,
                                  float eps, const AttnBwdPlan& p, float* ws, float* d_txt, float* d_enc, hipStream_t s) {
    using P = const typename In::T*;
    float *PT = ws, *DST = ws + p.pt_floats, *part = ws + 2 * p.pt_floats;
    hipLaunchKernelGGL((attn_fuse_bwd_words_kernel<In, T, FTM>), dim3(p.WT, B), dim3(64), sizeof(float) * 32 * (h + 4), s,
                       (P)vis, (P)txt, (P)vis_mid, (P)enc_x, gamma, dout, ld_dout_b, ld_dout_l, L, V, d, h, eps, PT, DST, p.Vp, p.Lp, part,
                       d_txt, d_enc);
}

}  // namespace vlg

extern "C" {

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


size_t vlg_attn_fuse_backward_workspace(int B, int L, int V, int h) {
    if (B < 1 || L < 1 || V < 1 || h < 1) return 0;
    return vlg::AttnBwdPlan(B, L, V, h).bytes;
}

int vlg_attn_fuse_backward(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                           const float* dout, long long ld_dout_b, long long ld_dout_l, int B, int L, int V, int d, int h, int in_dtype,
                           float eps, void* ws, size_t ws_bytes, float* d_vis, float* d_txt, float* d_vis_mid, float* d_enc_x, float* d_gamma,
                           float* d_beta, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || V < 1 || d < 1 || h < 1)
        return set_error(VLG_ERR_SHAPE, "attn_fuse_backward: bad shape B=%d L=%d V=%d d=%d h=%d", B, L, V, d, h);
    if (d % 16 || h % 16 || d > 16 * kAttnMaxFT || h > 16 * kAttnMaxCT)
        return set_error(VLG_ERR_SHAPE, "attn_fuse_backward: needs d, h multiples of 16 and <= 256 (got d=%d h=%d)", d, h);
    if ((size_t)V * h * 4 >= (1u << 31) || (size_t)V * d * 4 >= (1u << 31) || (size_t)L * h * 4 >= (1u << 31) ||
        (size_t)L * d * 4 >= (1u << 31))
        return set_error(VLG_ERR_SHAPE, "attn_fuse_backward: a sentence's rows exceed 2 GiB");
    if (!d_gamma || !d_beta) return set_error(VLG_ERR_ARG, "attn_fuse_backward: null output");
    if (ld_dout_b < 0 || ld_dout_l < 0 || ld_dout_b % 4 || ld_dout_l % 4)
        return set_error(VLG_ERR_SHAPE, "attn_fuse_backward: dout strides %lld, %lld (>= 0, multiples of 4 elements)", ld_dout_b, ld_dout_l);
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) {
        hipError_t e = hipMemsetAsync(d_gamma, 0, sizeof(float) * h, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_beta, 0, sizeof(float) * h, s);
        return e == hipSuccess ? 0 : set_error((int)e, "attn_fuse_backward: %s", hipGetErrorString(e));
    }
    if (!vis || !txt || !vis_mid || !enc_x || !gamma || !dout || !d_vis || !d_txt || !d_vis_mid || !d_enc_x)
        return set_error(VLG_ERR_ARG, "attn_fuse_backward: null buffer");
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "attn_fuse_backward: B=%d exceeds grid.y", B);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "attn_fuse_backward: in_dtype %d", in_dtype);
    const AttnBwdPlan p(B, L, V, h);
    if (!ws || ws_bytes < p.bytes)
        return set_error(VLG_ERR_WORKSPACE, "attn_fuse_backward: workspace %zu bytes < %zu", ws_bytes, p.bytes);
    float* wsf = (float*)ws;
#define VLG_BW(INV, TT)                                                                                            \
    do {                                                                                                           \
        if (d > 128) {                                                                                             \
            launch_attn_bwd_words<INV, TT, 16>(vis, txt, vis_mid, enc_x, gamma, dout, (size_t)ld_dout_b, (size_t)ld_dout_l, B, L, V, d, h, eps, p, wsf, d_txt, \
                                               d_enc_x, s);                                                        \
            hipLaunchKernelGGL((attn_fuse_bwd_regions_kernel<INV, 16>), dim3((V + 15) / 16, B), dim3(64), 0, s,    \
                               (const INV::T*)txt, d_enc_x, wsf, wsf + p.pt_floats, p.Vp, p.Lp, L, V, d, h, d_vis_mid, \
                               d_vis);                                                                             \
        } else {                                                                                                   \
            launch_attn_bwd_words<INV, TT, 8>(vis, txt, vis_mid, enc_x, gamma, dout, (size_t)ld_dout_b, (size_t)ld_dout_l, B, L, V, d, h, eps, p, wsf, d_txt, \
                                              d_enc_x, s);                                                         \
            hipLaunchKernelGGL((attn_fuse_bwd_regions_kernel<INV, 8>), dim3((V + 15) / 16, B), dim3(64), 0, s,     \
                               (const INV::T*)txt, d_enc_x, wsf, wsf + p.pt_floats, p.Vp, p.Lp, L, V, d, h, d_vis_mid, \
                               d_vis);                                                                             \
        }                                                                                                          \
    } while (0)
#define VLG_BWT(INV)                            \
    switch (p.T) {                              \
        case 1: VLG_BW(INV, 1); break;          \
        case 2: VLG_BW(INV, 2); break;          \
        case 3: VLG_BW(INV, 3); break;          \
        default: VLG_BW(INV, 4); break;         \
    }
    if (in_dtype == VLG_F32) { VLG_BWT(F32In) } else { VLG_BWT(BF16In) }
#undef VLG_BWT
#undef VLG_BW
    hipLaunchKernelGGL(attn_fuse_bwd_affine_kernel, dim3((2 * h + 63) / 64), dim3(1024), 0, s, wsf + 2 * p.pt_floats,
                       B * p.WT, h, d_gamma, d_beta);
    return check_launch("attn_fuse_backward");
}
}  // extern "C"
