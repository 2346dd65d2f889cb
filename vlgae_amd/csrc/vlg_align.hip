// vlg_align.hip -- region x word alignment kernels (gfx950) and their C-ABI entry points.
//
//   vlg_bilinear_align : DependencyBoxRel.gather_logit_simple  (src/model/joint.py:406-419)
//   vlg_attn_fuse      : the attention-fuse feeding the parser  (src/model/joint.py:670-674)
//
// v1: LDS-tiled fp32 FMA kernels with the mask / max-over-V / max-over-Q / batch-diagonal epilogue
// fused behind the contraction (the [B,A,Q,V] tensor is only written when the caller asks for it).
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
    const size_t lds = sizeof(float) * ((size_t)(kQT + kVT) * (d + 1) + (size_t)kQT * (kVT + 1) + Q + V);
    if (lds > 160 * 1024) return set_error(VLG_ERR_SHAPE, "bilinear_align: d=%d Q=%d V=%d exceed the LDS tile budget", d, Q, V);
    int a_per_block = 8;
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "bilinear_align: B=%d exceeds grid.y", B);
    dim3 grid((A + a_per_block - 1) / a_per_block, B);
    hipStream_t s = (hipStream_t)stream;
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
    if (in_dtype == VLG_F32) VLG_LAUNCH(F32In);
    else if (in_dtype == VLG_BF16) VLG_LAUNCH(BF16In);
    else return set_error(VLG_ERR_DTYPE, "bilinear_align: in_dtype %d", in_dtype);
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
    const size_t tile_f = (size_t)kFT * (size_t)((d + 1) > h ? (d + 1) : h);
    const size_t per_q = (size_t)(d + 1) + V + h + 2;
    int QC = L < 32 ? L : 32;
    while (QC > 1 && sizeof(float) * (tile_f + per_q * QC) > 150 * 1024) QC >>= 1;
    const size_t lds = sizeof(float) * (tile_f + per_q * QC);
    if (lds > 150 * 1024) return set_error(VLG_ERR_SHAPE, "attn_fuse: V=%d d=%d h=%d exceed the LDS budget", V, d, h);
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "attn_fuse: B=%d exceeds grid.y", B);
    hipStream_t s = (hipStream_t)stream;
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
    else if (in_dtype == VLG_BF16) VLG_LAUNCH(BF16In);
    else return set_error(VLG_ERR_DTYPE, "attn_fuse: in_dtype %d", in_dtype);
#undef VLG_LAUNCH
    return check_launch("attn_fuse_kernel");
}

}  // extern "C"
