// vlg_langfeat.hip -- the byte work around the language-side encoders of lang_feat_max_tree (src/model/joint.py:235-292):
// everything between the DP outputs / the fused encodings and the matrix-core kernels that is not a GEMM.  Each kernel
// replaces a handful of torch launches (cat / masked_fill / sum / div / gather / LeakyReLU / cast) with one HBM-bound pass.
//
//   langfeat_x1_kernel         joint.py:262-266   root = masked mean of the word encodings; x = cat([root, x])      -> bf16 [B,N,h]
//   langfeat_x1_bwd_kernel                         its adjoint (the mean spreads the root row's gradient over the words)
//   langfeat_split_kernel      joint.py:267-273   word / child / parent encoders' epilogue on the fused projection
//                                                  [M, 3d] = x W_cat^T + b_cat: LeakyReLU on the child / parent thirds
//                                                  (MLP, nn/common.py:47-51), parent rows gathered by the predicted heads
//                                                  (x.gather(1, predicted...) commutes with the row-wise encoder),
//                                                  word rows written straight into the first half of txt [B,2N,d] (:288)
//   langfeat_split_bwd_kernel                      its adjoint: LeakyReLU', scatter-add of the parent rows by head (fixed
//                                                  order: one thread owns a column of a sentence), one [M,3d] cotangent
//   langfeat_marginal_kernel   joint.py:246-261   arc_margin = grad.sum(-1).gather(-1, predicted); txt_marginal =
//                                                  cat([mask, arc_margin]); txt_mask = cat([mask, mask])
//   langfeat_arc_out_kernel    joint.py:278-288   arc_repr = trilinear + affine, cast, into the second half of txt
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"

namespace vlg {

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) {   // round to nearest even, like torch's cast
    const uint32_t u = __float_as_uint(v);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float ldf(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const uint16_t* p, size_t i) { return bf2f(p[i]); }

// x [B,L,h] (T) -> x1 [B,L+1,h] bf16: row 0 = sum_{l < len} x[b,l] / len, rows 1.. = x.  grid = (B, ceil(h / 256)).
template <typename T>
__global__ __launch_bounds__(256) void langfeat_x1_kernel(const T* __restrict__ x, const int64_t* __restrict__ lengths, int L,
                                                          int h, uint16_t* __restrict__ x1) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= h) return;
    const int len = min(max((int)lengths[b], 0), L);
    const T* src = x + (size_t)b * L * h + c;
    uint16_t* dst = x1 + (size_t)b * (L + 1) * h + c;
    float acc = 0.f;
#pragma unroll 4
    for (int l = 0; l < L; ++l) {
        const float v = ldf(src, (size_t)l * h);
        if (l < len) acc += v;
        dst[(size_t)(l + 1) * h] = f2bf(v);
    }
    dst[0] = f2bf(acc / (float)max(len, 1));
}

// d_x1 [B,L+1,h] (T) -> d_x [B,L,h] fp32: d_x[b,l] = d_x1[b,l+1] + (l < len) d_x1[b,0] / len
template <typename T>
__global__ __launch_bounds__(256) void langfeat_x1_bwd_kernel(const T* __restrict__ d_x1, const int64_t* __restrict__ lengths,
                                                              int L, int h, float* __restrict__ d_x) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= h) return;
    const int len = min(max((int)lengths[b], 0), L);
    const T* src = d_x1 + (size_t)b * (L + 1) * h + c;
    float* dst = d_x + (size_t)b * L * h + c;
    const float share = ldf(src, 0) / (float)max(len, 1);
#pragma unroll 4
    for (int l = 0; l < L; ++l) dst[(size_t)l * h] = ldf(src, (size_t)(l + 1) * h) + (l < len ? share : 0.f);
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// pre [B*N, 3d] bf16 -> txt[b, n, :] = pre[:, 0:d];  child[m] = leaky(pre[m, d:2d]);  parent[b,n] = leaky(pre[b, heads[b,n], 2d:3d]);
// sum[m] = child + parent (bf16 add of the rounded values, like torch's).  One thread = 8 channels of one row.
__global__ __launch_bounds__(256) void langfeat_split_kernel(const uint16_t* __restrict__ pre, const int64_t* __restrict__ heads,
                                                             int B, int N, int d, float slope, uint16_t* __restrict__ txt,
                                                             uint16_t* __restrict__ child, uint16_t* __restrict__ parent,
                                                             uint16_t* __restrict__ sum) {
    const int per_row = d >> 3;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t m = t / per_row;
    if (m >= (size_t)B * N) return;
    const int c8 = (int)(t - m * per_row) * 8;
    const int b = (int)(m / N), n = (int)(m - (size_t)b * N);
    int hd = (int)heads[m];
    hd = min(max(hd, 0), N - 1);
    const uint16_t* row = pre + m * 3 * d;
    const uint16_t* prow = pre + ((size_t)b * N + hd) * 3 * d;
    const uint4 w = *reinterpret_cast<const uint4*>(row + c8);
    *reinterpret_cast<uint4*>(txt + ((size_t)b * 2 * N + n) * d + c8) = w;
    const uint4 cv = *reinterpret_cast<const uint4*>(row + d + c8);
    const uint4 pv = *reinterpret_cast<const uint4*>(prow + 2 * d + c8);
    const uint16_t* cs = reinterpret_cast<const uint16_t*>(&cv);
    const uint16_t* ps = reinterpret_cast<const uint16_t*>(&pv);
    uint4 co, po, so;
    uint16_t* cop = reinterpret_cast<uint16_t*>(&co);
    uint16_t* pop = reinterpret_cast<uint16_t*>(&po);
    uint16_t* sop = reinterpret_cast<uint16_t*>(&so);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        cop[i] = f2bf(leaky(bf2f(cs[i]), slope));
        pop[i] = f2bf(leaky(bf2f(ps[i]), slope));
        sop[i] = f2bf(bf2f(cop[i]) + bf2f(pop[i]));
    }
    *reinterpret_cast<uint4*>(child + m * d + c8) = co;
    *reinterpret_cast<uint4*>(parent + m * d + c8) = po;
    if (sum) *reinterpret_cast<uint4*>(sum + m * d + c8) = so;
}

// d_pre [B*N, 3d] bf16 from: d_word = d_txt[b, n, :] (T1; row stride given), d_child / d_parent fp32 [B*N, d] (+ optional
// extra cotangent of the sum, fp32, added to both), child / parent bf16 (the activations: their sign is LeakyReLU's branch).
// grid = B.  The scatter-add over the children of a head is a gather over the sentence's rows in ascending order (fixed
// summation order, no atomics).  LDS: [N][d] fp32 + N ints.
constexpr int kSplitBwdThreads = 1024;
template <typename T1, typename T2>
__global__ __launch_bounds__(kSplitBwdThreads) void langfeat_split_bwd_kernel(
    const T1* __restrict__ d_txt, const float* __restrict__ d_child, const float* __restrict__ d_parent,
    const T2* __restrict__ d_sum, const uint16_t* __restrict__ child, const uint16_t* __restrict__ parent,
    const int64_t* __restrict__ heads, int N, int d, float slope, uint16_t* __restrict__ d_pre) {
    extern __shared__ float gbuf[];   // [N][d] parent-third cotangents before the scatter, then N head indices + their CSR by head
    int* hd = reinterpret_cast<int*>(gbuf + N * d);
    int* order = hd + N;        // children sorted by (head, position)
    int* start = order + N;     // start[j] .. start[j+1]: the children of head j in `order`
    const int b = blockIdx.x;
    const size_t m0 = (size_t)b * N;
    for (int n = threadIdx.x; n < N; n += kSplitBwdThreads) hd[n] = min(max((int)heads[m0 + n], 0), N - 1);
    // element-wise part, coalesced over channels; the loads of U items are issued together (one workgroup per CU: the
    // latency of a dependent load per item is what this pass would otherwise cost)
    constexpr int U = 3;
    for (int i0 = threadIdx.x; i0 < N * d; i0 += U * kSplitBwdThreads) {
        float gw[U], gc[U], gp[U], ex[U], cv[U], pv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * kSplitBwdThreads, N * d - 1);
            const int n = i / d, c = i - n * d;
            const size_t m = m0 + n;
            ex[u] = d_sum ? ldf(d_sum, m * d + c) : 0.f;
            gw[u] = ldf(d_txt, ((size_t)b * 2 * N + n) * d + c);
            gc[u] = d_child[m * d + c];
            gp[u] = d_parent[m * d + c];
            cv[u] = bf2f(child[m * d + c]);
            pv[u] = bf2f(parent[m * d + c]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSplitBwdThreads;
            if (i < N * d) {
                const int n = i / d, c = i - n * d;
                gbuf[i] = (gp[u] + ex[u]) * (pv[u] > 0.f ? 1.f : slope);
                uint16_t* o = d_pre + (m0 + n) * 3 * d;
                o[c] = f2bf(gw[u]);
                o[d + c] = f2bf((gc[u] + ex[u]) * (cv[u] > 0.f ? 1.f : slope));
            }
        }
    }
    __syncthreads();   // hd (and gbuf) complete
    // counting sort of the N children by head (N <= ~100: one thread per child / per head, N compares each)
    for (int n = threadIdx.x; n <= N; n += kSplitBwdThreads) {
        int lower = 0, rank = 0;
        const int hn = n < N ? hd[n] : 0;
        for (int k = 0; k < N; ++k) {
            const int hk = hd[k];
            lower += hk < n;                                   // children of heads below n (n read as a head index here)
            rank += (hk < hn) || (hk == hn && k < n);
        }
        start[n] = lower;
        if (n < N) order[rank] = n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N * d; i += kSplitBwdThreads) {   // row j of the parent third = sum of its children's rows, n ascending
        const int j = i / d, c = i - j * d;
        float s = 0.f;
        for (int k = start[j]; k < start[j + 1]; ++k) s += gbuf[order[k] * d + c];
        d_pre[(m0 + j) * 3 * d + 2 * d + c] = f2bf(s);
    }
}

// gatt [B,N,N,2] fp32 (d logZ / d attach), heads [B,N], lengths [B] -> txt_marginal [B,2N] fp32, txt_mask [B,2N] u8
__global__ __launch_bounds__(256) void langfeat_marginal_kernel(const float* __restrict__ gatt, const int64_t* __restrict__ heads,
                                                                const int64_t* __restrict__ lengths, int B, int N, int use_marginal,
                                                                float* __restrict__ marg, uint8_t* __restrict__ mask) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * N) return;
    const int b = i / N, n = i - b * N;
    const int len = (int)lengths[b];
    const float mk = (n >= 1 && n <= len) ? 1.f : 0.f;           // mask = cat([zeros(B,1), vp.mask]), joint.py:248
    float am = mk;
    if (use_marginal) {                                          // arc_margin.gather(-1, predicted): [b, n, predicted[b,n]], :258-260
        const int hd = min(max((int)heads[i], 0), N - 1);
        const float2 g = *reinterpret_cast<const float2*>(gatt + (((size_t)b * N + n) * N + hd) * 2);
        am = g.x + g.y;
    }
    marg[(size_t)b * 2 * N + n] = mk;
    marg[(size_t)b * 2 * N + N + n] = am;
    mask[(size_t)b * 2 * N + n] = mk != 0.f;
    mask[(size_t)b * 2 * N + N + n] = mk != 0.f;
}

// txt[b, N + n, :] = bf16(tri[m,:] + aff[m,:])   (aff: bf16 [M,d] = (child + parent) w2 + b from the library GEMM, or null)
__global__ __launch_bounds__(256) void langfeat_arc_out_kernel(const float* __restrict__ tri, const uint16_t* __restrict__ aff, int B,
                                                               int N, int d, uint16_t* __restrict__ txt) {
    const int per_row = d >> 2;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t m = t / per_row;
    if (m >= (size_t)B * N) return;
    const int c4 = (int)(t - m * per_row) * 4;
    const int b = (int)(m / N), n = (int)(m - (size_t)b * N);
    float4 v = *reinterpret_cast<const float4*>(tri + m * d + c4);
    if (aff) {
        const uint2 a = *reinterpret_cast<const uint2*>(aff + m * d + c4);
        v.x += __uint_as_float(a.x << 16); v.y += __uint_as_float(a.x & 0xffff0000u);
        v.z += __uint_as_float(a.y << 16); v.w += __uint_as_float(a.y & 0xffff0000u);
    }
    uint2 o;
    o.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
    o.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
    *reinterpret_cast<uint2*>(txt + ((size_t)b * 2 * N + N + n) * d + c4) = o;
}

}  // namespace

}  // namespace vlg

extern "C" {

int vlg_langfeat_root_cat(const void* x, const int64_t* lengths, int B, int L, int h, int in_dtype, void* x1, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || h < 1) return set_error(VLG_ERR_SHAPE, "langfeat_root_cat: bad shape B=%d L=%d h=%d", B, L, h);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "langfeat_root_cat: in_dtype %d", in_dtype);
    if (B == 0) return 0;
    if (!x || !lengths || !x1) return set_error(VLG_ERR_ARG, "langfeat_root_cat: null buffer");
    const dim3 grid(B, (h + 255) / 256);
    if (in_dtype == VLG_F32)
        hipLaunchKernelGGL(langfeat_x1_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)x, lengths, L, h, (uint16_t*)x1);
    else
        hipLaunchKernelGGL(langfeat_x1_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, lengths, L, h, (uint16_t*)x1);
    return check_launch("langfeat_x1_kernel");
}

int vlg_langfeat_root_cat_backward(const void* d_x1, const int64_t* lengths, int B, int L, int h, int in_dtype, float* d_x,
                                   void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || h < 1) return set_error(VLG_ERR_SHAPE, "langfeat_root_cat_backward: bad shape B=%d L=%d h=%d", B, L, h);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "langfeat_root_cat_backward: in_dtype %d", in_dtype);
    if (B == 0) return 0;
    if (!d_x1 || !lengths || !d_x) return set_error(VLG_ERR_ARG, "langfeat_root_cat_backward: null buffer");
    const dim3 grid(B, (h + 255) / 256);
    if (in_dtype == VLG_F32)
        hipLaunchKernelGGL(langfeat_x1_bwd_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)d_x1, lengths, L, h, d_x);
    else
        hipLaunchKernelGGL(langfeat_x1_bwd_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d_x1, lengths, L, h, d_x);
    return check_launch("langfeat_x1_bwd_kernel");
}

int vlg_langfeat_split(const void* pre, const int64_t* heads, int B, int N, int d, float slope, void* txt, void* child,
                       void* parent, void* sum, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2 || d < 8 || d % 8) return set_error(VLG_ERR_SHAPE, "langfeat_split: bad shape B=%d N=%d d=%d (d a multiple of 8)", B, N, d);
    if (B == 0) return 0;
    if (!pre || !heads || !txt || !child || !parent) return set_error(VLG_ERR_ARG, "langfeat_split: null buffer");
    const size_t threads = (size_t)B * N * (d / 8);
    hipLaunchKernelGGL(langfeat_split_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t*)pre, heads, B, N, d, slope, (uint16_t*)txt, (uint16_t*)child, (uint16_t*)parent, (uint16_t*)sum);
    return check_launch("langfeat_split_kernel");
}

int vlg_langfeat_split_backward(const void* d_txt, int d_txt_dtype, const float* d_child, const float* d_parent, const void* d_sum,
                                int d_sum_dtype, const void* child, const void* parent, const int64_t* heads, int B, int N, int d,
                                float slope, void* d_pre, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2 || d < 1) return set_error(VLG_ERR_SHAPE, "langfeat_split_backward: bad shape B=%d N=%d d=%d", B, N, d);
    if ((d_txt_dtype != VLG_F32 && d_txt_dtype != VLG_BF16) || (d_sum && d_sum_dtype != VLG_F32 && d_sum_dtype != VLG_BF16))
        return set_error(VLG_ERR_DTYPE, "langfeat_split_backward: dtypes %d / %d", d_txt_dtype, d_sum_dtype);
    if ((size_t)(N * (d + 3) + 1) * sizeof(float) > 64 * 1024) return set_error(VLG_ERR_SHAPE, "langfeat_split_backward: N*d = %d exceeds the 64 KB LDS tile", N * d);
    if (B == 0) return 0;
    if (!d_txt || !d_child || !d_parent || !child || !parent || !heads || !d_pre) return set_error(VLG_ERR_ARG, "langfeat_split_backward: null buffer");
    const size_t lds = sizeof(float) * (size_t)(N * (d + 3) + 1);
    const uint16_t *c = (const uint16_t*)child, *p = (const uint16_t*)parent;
    uint16_t* o = (uint16_t*)d_pre;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(B), block(kSplitBwdThreads);
    const bool t32 = d_txt_dtype == VLG_F32, s32 = !d_sum || d_sum_dtype == VLG_F32;
    if (t32 && s32)
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<float, float>), grid, block, lds, s, (const float*)d_txt, d_child, d_parent, (const float*)d_sum, c, p, heads, N, d, slope, o);
    else if (t32)
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<float, uint16_t>), grid, block, lds, s, (const float*)d_txt, d_child, d_parent, (const uint16_t*)d_sum, c, p, heads, N, d, slope, o);
    else if (s32)
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<uint16_t, float>), grid, block, lds, s, (const uint16_t*)d_txt, d_child, d_parent, (const float*)d_sum, c, p, heads, N, d, slope, o);
    else
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<uint16_t, uint16_t>), grid, block, lds, s, (const uint16_t*)d_txt, d_child, d_parent, (const uint16_t*)d_sum, c, p, heads, N, d, slope, o);
    return check_launch("langfeat_split_bwd_kernel");
}

int vlg_langfeat_marginal(const float* grad_attach, const int64_t* heads, const int64_t* lengths, int B, int N, int use_marginal,
                          float* txt_marginal, uint8_t* txt_mask, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2) return set_error(VLG_ERR_SHAPE, "langfeat_marginal: bad shape B=%d N=%d", B, N);
    if (B == 0) return 0;
    if (!lengths || !txt_marginal || !txt_mask || (use_marginal && (!grad_attach || !heads)))
        return set_error(VLG_ERR_ARG, "langfeat_marginal: null buffer");
    hipLaunchKernelGGL(langfeat_marginal_kernel, dim3((B * N + 255) / 256), dim3(256), 0, (hipStream_t)stream, grad_attach, heads, lengths,
                       B, N, use_marginal, txt_marginal, txt_mask);
    return check_launch("langfeat_marginal_kernel");
}

int vlg_langfeat_arc_out(const float* tri, const void* aff, int B, int N, int d, void* txt, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2 || d < 4 || d % 4) return set_error(VLG_ERR_SHAPE, "langfeat_arc_out: bad shape B=%d N=%d d=%d (d a multiple of 4)", B, N, d);
    if (B == 0) return 0;
    if (!tri || !txt) return set_error(VLG_ERR_ARG, "langfeat_arc_out: null buffer");
    const size_t threads = (size_t)B * N * (d / 4);
    hipLaunchKernelGGL(langfeat_arc_out_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tri,
                       (const uint16_t*)aff, B, N, d, (uint16_t*)txt);
    return check_launch("langfeat_arc_out_kernel");
}

}  // extern "C"
