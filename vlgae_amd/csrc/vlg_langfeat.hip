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
// Activations (x1, pre, txt, child, parent, sum, d_pre) are stored as bf16 (BASELINE.json's configs) or fp32 (the reference's
// `precision: 32`, config/trainer/train.yaml:20): every kernel is a template over the storage type A; arithmetic is fp32.
// SharedDropout (nn/dropout.py:42-63; word / child / parent encoders, p = 0.33 in config/model/vlgae.yaml:69-73) multiplies an
// encoder's output by a mask [B,1,d] shared over the positions of a sentence, AFTER the activation (nn/common.py:47-51): the
// caller passes the three masks as drop [B,3,d] fp32 (0 or 1/(1-p)); NULL = identity (eval).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_rows.h"

namespace vlg {

namespace {

// x [B,L,h] (T) -> x1 [B,L+1,h] bf16: row 0 = sum_{l < len} x[b,l] / len, rows 1.. = x.  grid = (B, ceil(h / 256)).
template <typename T, typename A>
__global__ __launch_bounds__(256) void langfeat_x1_kernel(const T* __restrict__ x, const int64_t* __restrict__ lengths, int L,
                                                          int h, A* __restrict__ x1) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= h) return;
    const int len = min(max((int)lengths[b], 0), L);
    const T* src = x + (size_t)b * L * h + c;
    A* dst = x1 + (size_t)b * (L + 1) * h + c;
    float acc = 0.f;
#pragma unroll 4
    for (int l = 0; l < L; ++l) {
        const float v = ldf(src, (size_t)l * h);
        if (l < len) acc += v;
        stf(dst, (size_t)(l + 1) * h, v);
    }
    stf(dst, 0, acc / (float)max(len, 1));
}

// d_x1 [B,L+1,h] (T) -> d_x [B,L,h] fp32: d_x[b,l] = d_x1[b,l+1] + (l < len) d_x1[b,0] / len
template <typename T, typename O>
__global__ __launch_bounds__(256) void langfeat_x1_bwd_kernel(const T* __restrict__ d_x1, const int64_t* __restrict__ lengths,
                                                              int L, int h, O* __restrict__ d_x) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= h) return;
    const int len = min(max((int)lengths[b], 0), L);
    const T* src = d_x1 + (size_t)b * (L + 1) * h + c;
    O* dst = d_x + (size_t)b * L * h + c;
    const float share = ldf(src, 0) / (float)max(len, 1);
#pragma unroll 4
    for (int l = 0; l < L; ++l) stf(dst, (size_t)l * h, ldf(src, (size_t)(l + 1) * h) + (l < len ? share : 0.f));
}

// pre [B*N, 3d] (A) -> txt[b, n, :] = pre[:, 0:d] * m_word;  child[m] = leaky(pre[m, d:2d]) * m_child;
// parent[b,n] = leaky(pre[b, heads[b,n], 2d:3d]) * m_parent;  sum[m] = child + parent (the add of the STORED values, like torch's).
// drop [B,3,d] fp32 = the three SharedDropout masks of sentence b (NULL: ones).  One thread = 8 channels of one row.
template <typename A>
__global__ __launch_bounds__(256) void langfeat_split_kernel(const A* __restrict__ pre, const int64_t* __restrict__ heads,
                                                             const float* __restrict__ drop, int ld_drop, int B, int N, int d, float slope,
                                                             A* __restrict__ txt, A* __restrict__ child, A* __restrict__ parent,
                                                             A* __restrict__ sum) {
    const int per_row = d >> 3;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t m = t / per_row;
    if (m >= (size_t)B * N) return;
    const int c8 = (int)(t - m * per_row) * 8;
    const int b = (int)(m / N), n = (int)(m - (size_t)b * N);
    int hd = (int)heads[m];
    hd = min(max(hd, 0), N - 1);
    const A* row = pre + m * 3 * d;
    const A* prow = pre + ((size_t)b * N + hd) * 3 * d;
    float w[8], cv[8], pv[8], mw[8], mc[8], mp[8];
    load8(row + c8, w);
    load8(row + d + c8, cv);
    load8(prow + 2 * d + c8, pv);
    if (drop) {
        const float* dr = drop + (size_t)b * ld_drop + c8;
        load8(dr, mw); load8(dr + d, mc); load8(dr + 2 * d, mp);
    }
    float so[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        cv[i] = leaky(cv[i], slope);
        pv[i] = leaky(pv[i], slope);
        if (drop) { w[i] *= mw[i]; cv[i] *= mc[i]; pv[i] *= mp[i]; }
        so[i] = stored<A>(cv[i]) + stored<A>(pv[i]);
    }
    store8(txt + ((size_t)b * 2 * N + n) * d + c8, w);
    store8(child + m * d + c8, cv);
    store8(parent + m * d + c8, pv);
    if (sum) store8(sum + m * d + c8, so);
}

// d_pre [B*N, 3d] (A) from: d_word = d_txt[b, n, :] (T1; row stride given), d_child / d_parent fp32 [B*N, d] (+ optional
// extra cotangent of the sum, fp32, added to both), child / parent (A; the activations: their sign is LeakyReLU's branch -- with
// dropout a dropped channel's activation is 0 and so is its mask, whatever branch is read), drop [B,3,d] fp32 or NULL.
// grid = B.  The scatter-add over the children of a head is a gather over the sentence's rows in ascending order (fixed
// summation order, no atomics).  LDS: [N][d] fp32 + N ints.
constexpr int kSplitBwdThreads = 1024;
template <typename T1, typename T2, typename A>
__global__ __launch_bounds__(kSplitBwdThreads) void langfeat_split_bwd_kernel(
    const T1* __restrict__ d_txt, const float* __restrict__ d_child, const float* __restrict__ d_parent,
    const T2* __restrict__ d_sum, const A* __restrict__ child, const A* __restrict__ parent,
    const int64_t* __restrict__ heads, const float* __restrict__ drop, int ld_drop, int N, int d, float slope, A* __restrict__ d_pre) {
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
        float gw[U], gc[U], gp[U], ex[U], cv[U], pv[U], mw[U], mc[U], mp[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = min(i0 + u * kSplitBwdThreads, N * d - 1);
            const int n = i / d, c = i - n * d;
            const size_t m = m0 + n;
            ex[u] = d_sum ? ldf(d_sum, m * d + c) : 0.f;
            gw[u] = ldf(d_txt, ((size_t)b * 2 * N + n) * d + c);
            gc[u] = d_child[m * d + c];
            gp[u] = d_parent[m * d + c];
            cv[u] = ldf(child, m * d + c);
            pv[u] = ldf(parent, m * d + c);
            mw[u] = drop ? drop[(size_t)b * ld_drop + c] : 1.f;
            mc[u] = drop ? drop[(size_t)b * ld_drop + d + c] : 1.f;
            mp[u] = drop ? drop[(size_t)b * ld_drop + 2 * d + c] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * kSplitBwdThreads;
            if (i < N * d) {
                const int n = i / d, c = i - n * d;
                gbuf[i] = (gp[u] + ex[u]) * mp[u] * (pv[u] > 0.f ? 1.f : slope);
                A* o = d_pre + (m0 + n) * 3 * d;
                stf(o, c, gw[u] * mw[u]);
                stf(o, d + c, (gc[u] + ex[u]) * mc[u] * (cv[u] > 0.f ? 1.f : slope));
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
        stf(d_pre, (m0 + j) * 3 * d + 2 * d + c, s);
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

// txt[b, N + n, :] = A(tri[m,:] + aff[m,:])   (aff: A [M,d] = (child + parent) w2 + b from the library GEMM, or null)
template <typename A>
__global__ __launch_bounds__(256) void langfeat_arc_out_kernel(const float* __restrict__ tri, const A* __restrict__ aff, int B,
                                                               int N, int d, A* __restrict__ txt) {
    const int per_row = d >> 3;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t m = t / per_row;
    if (m >= (size_t)B * N) return;
    const int c8 = (int)(t - m * per_row) * 8;
    const int b = (int)(m / N), n = (int)(m - (size_t)b * N);
    float v[8], a[8];
    load8(tri + m * d + c8, v);
    if (aff) {
        load8(aff + m * d + c8, a);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] += a[i];
    }
    store8(txt + ((size_t)b * 2 * N + N + n) * d + c8, v);
}

// out[b, n, c] = pre[b, n, c] * drop[b, c] for c < d, 0 for d <= c < width  (word-only features, joint.py:193-211 in training mode; drop null =
// identity).  Rows ld_in / ld_out elements apart: the word third of the three encoders' shared projection [M, 3d] is read where it lies, and the
// adjoint writes the full-width cotangent of that projection (zeros for the child | parent thirds) in one pass.  In place allowed (equal strides).
template <typename A>
__global__ __launch_bounds__(256) void langfeat_rowscale_kernel(const A* __restrict__ pre, int ld_in, const float* __restrict__ drop, int B, int N,
                                                                int d, int ld_drop, A* __restrict__ out, int ld_out, int width) {
    const int per_row = width >> 3;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t m = t / per_row;
    if (m >= (size_t)B * N) return;
    const int c8 = (int)(t - m * per_row) * 8;
    const int b = (int)(m / N);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (c8 < d) {
        load8(pre + m * ld_in + c8, v);
        if (drop) {
            float k[8];
            load8(drop + (size_t)b * ld_drop + c8, k);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= k[i];
        }
    }
    store8(out + m * ld_out + c8, v);
}

}  // namespace

}  // namespace vlg

namespace {
inline bool ok_dtype(int t) { return t == VLG_F32 || t == VLG_BF16; }
}

namespace {
template <typename A>
void launch_split_bwd(const void* d_txt, bool t32, const float* d_child, const float* d_parent, const void* d_sum, bool s32,
                      const void* child, const void* parent, const int64_t* heads, const float* drop, int ld_drop, int B, int N, int d, float slope,
                      void* d_pre, hipStream_t s) {
    using namespace vlg;
    const size_t lds = sizeof(float) * (size_t)(N * (d + 3) + 1);
    const A *c = (const A*)child, *p = (const A*)parent;
    A* o = (A*)d_pre;
    const dim3 grid(B), block(kSplitBwdThreads);
    if (t32 && s32)
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<float, float, A>), grid, block, lds, s, (const float*)d_txt, d_child, d_parent, (const float*)d_sum, c, p, heads, drop, ld_drop, N, d, slope, o);
    else if (t32)
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<float, uint16_t, A>), grid, block, lds, s, (const float*)d_txt, d_child, d_parent, (const uint16_t*)d_sum, c, p, heads, drop, ld_drop, N, d, slope, o);
    else if (s32)
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<uint16_t, float, A>), grid, block, lds, s, (const uint16_t*)d_txt, d_child, d_parent, (const float*)d_sum, c, p, heads, drop, ld_drop, N, d, slope, o);
    else
        hipLaunchKernelGGL((langfeat_split_bwd_kernel<uint16_t, uint16_t, A>), grid, block, lds, s, (const uint16_t*)d_txt, d_child, d_parent, (const uint16_t*)d_sum, c, p, heads, drop, ld_drop, N, d, slope, o);
}
}  // namespace

extern "C" {

int vlg_langfeat_root_cat(const void* x, const int64_t* lengths, int B, int L, int h, int in_dtype, void* x1, int out_dtype,
                          void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || h < 1) return set_error(VLG_ERR_SHAPE, "langfeat_root_cat: bad shape B=%d L=%d h=%d", B, L, h);
    if (!ok_dtype(in_dtype) || !ok_dtype(out_dtype)) return set_error(VLG_ERR_DTYPE, "langfeat_root_cat: dtypes %d -> %d", in_dtype, out_dtype);
    if (B == 0) return 0;
    if (!x || !lengths || !x1) return set_error(VLG_ERR_ARG, "langfeat_root_cat: null buffer");
    const dim3 grid(B, (h + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == VLG_F32 && out_dtype == VLG_F32)
        hipLaunchKernelGGL((langfeat_x1_kernel<float, float>), grid, dim3(256), 0, s, (const float*)x, lengths, L, h, (float*)x1);
    else if (in_dtype == VLG_F32)
        hipLaunchKernelGGL((langfeat_x1_kernel<float, uint16_t>), grid, dim3(256), 0, s, (const float*)x, lengths, L, h, (uint16_t*)x1);
    else if (out_dtype == VLG_F32)
        hipLaunchKernelGGL((langfeat_x1_kernel<uint16_t, float>), grid, dim3(256), 0, s, (const uint16_t*)x, lengths, L, h, (float*)x1);
    else
        hipLaunchKernelGGL((langfeat_x1_kernel<uint16_t, uint16_t>), grid, dim3(256), 0, s, (const uint16_t*)x, lengths, L, h, (uint16_t*)x1);
    return check_launch("langfeat_x1_kernel");
}

int vlg_langfeat_root_cat_backward(const void* d_x1, const int64_t* lengths, int B, int L, int h, int in_dtype, void* d_x, int out_dtype,
                                   void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || h < 1) return set_error(VLG_ERR_SHAPE, "langfeat_root_cat_backward: bad shape B=%d L=%d h=%d", B, L, h);
    if (!ok_dtype(in_dtype) || !ok_dtype(out_dtype)) return set_error(VLG_ERR_DTYPE, "langfeat_root_cat_backward: dtypes %d -> %d", in_dtype, out_dtype);
    if (B == 0) return 0;
    if (!d_x1 || !lengths || !d_x) return set_error(VLG_ERR_ARG, "langfeat_root_cat_backward: null buffer");
    const dim3 grid(B, (h + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == VLG_F32 && out_dtype == VLG_F32)
        hipLaunchKernelGGL((langfeat_x1_bwd_kernel<float, float>), grid, dim3(256), 0, s, (const float*)d_x1, lengths, L, h, (float*)d_x);
    else if (in_dtype == VLG_F32)
        hipLaunchKernelGGL((langfeat_x1_bwd_kernel<float, uint16_t>), grid, dim3(256), 0, s, (const float*)d_x1, lengths, L, h, (uint16_t*)d_x);
    else if (out_dtype == VLG_F32)
        hipLaunchKernelGGL((langfeat_x1_bwd_kernel<uint16_t, float>), grid, dim3(256), 0, s, (const uint16_t*)d_x1, lengths, L, h, (float*)d_x);
    else
        hipLaunchKernelGGL((langfeat_x1_bwd_kernel<uint16_t, uint16_t>), grid, dim3(256), 0, s, (const uint16_t*)d_x1, lengths, L, h, (uint16_t*)d_x);
    return check_launch("langfeat_x1_bwd_kernel");
}

int vlg_langfeat_split(const void* pre, const int64_t* heads, const float* drop, int ld_drop, int B, int N, int d, int act_dtype, float slope,
                       void* txt, void* child, void* parent, void* sum, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2 || d < 8 || d % 8) return set_error(VLG_ERR_SHAPE, "langfeat_split: bad shape B=%d N=%d d=%d (d a multiple of 8)", B, N, d);
    if (!ok_dtype(act_dtype)) return set_error(VLG_ERR_DTYPE, "langfeat_split: act_dtype %d", act_dtype);
    if (B == 0) return 0;
    if (!pre || !heads || !txt || !child || !parent) return set_error(VLG_ERR_ARG, "langfeat_split: null buffer");
    if (drop && (ld_drop < 3 * d || ld_drop % 4)) return set_error(VLG_ERR_SHAPE, "langfeat_split: ld_drop=%d (>= 3d, a multiple of 4)", ld_drop);
    const size_t threads = (size_t)B * N * (d / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (act_dtype == VLG_F32)
        hipLaunchKernelGGL(langfeat_split_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)pre, heads, drop, ld_drop, B, N, d,
                           slope, (float*)txt, (float*)child, (float*)parent, (float*)sum);
    else
        hipLaunchKernelGGL(langfeat_split_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)pre, heads, drop, ld_drop, B, N,
                           d, slope, (uint16_t*)txt, (uint16_t*)child, (uint16_t*)parent, (uint16_t*)sum);
    return check_launch("langfeat_split_kernel");
}

int vlg_langfeat_split_backward(const void* d_txt, int d_txt_dtype, const float* d_child, const float* d_parent, const void* d_sum,
                                int d_sum_dtype, const void* child, const void* parent, const int64_t* heads, const float* drop, int ld_drop,
                                int B, int N, int d, int act_dtype, float slope, void* d_pre, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2 || d < 1) return set_error(VLG_ERR_SHAPE, "langfeat_split_backward: bad shape B=%d N=%d d=%d", B, N, d);
    if (!ok_dtype(d_txt_dtype) || (d_sum && !ok_dtype(d_sum_dtype)) || !ok_dtype(act_dtype))
        return set_error(VLG_ERR_DTYPE, "langfeat_split_backward: dtypes %d / %d / %d", d_txt_dtype, d_sum_dtype, act_dtype);
    if ((size_t)(N * (d + 3) + 1) * sizeof(float) > 64 * 1024) return set_error(VLG_ERR_SHAPE, "langfeat_split_backward: N*d = %d exceeds the 64 KB LDS tile", N * d);
    if (B == 0) return 0;
    if (!d_txt || !d_child || !d_parent || !child || !parent || !heads || !d_pre) return set_error(VLG_ERR_ARG, "langfeat_split_backward: null buffer");
    if (drop && ld_drop < 3 * d) return set_error(VLG_ERR_SHAPE, "langfeat_split_backward: ld_drop=%d < 3d", ld_drop);
    const bool t32 = d_txt_dtype == VLG_F32, s32 = !d_sum || d_sum_dtype == VLG_F32;
    if (act_dtype == VLG_F32)
        launch_split_bwd<float>(d_txt, t32, d_child, d_parent, d_sum, s32, child, parent, heads, drop, ld_drop, B, N, d, slope, d_pre, (hipStream_t)stream);
    else
        launch_split_bwd<uint16_t>(d_txt, t32, d_child, d_parent, d_sum, s32, child, parent, heads, drop, ld_drop, B, N, d, slope, d_pre, (hipStream_t)stream);
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

int vlg_langfeat_arc_out(const float* tri, const void* aff, int B, int N, int d, int act_dtype, void* txt, void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2 || d < 8 || d % 8) return set_error(VLG_ERR_SHAPE, "langfeat_arc_out: bad shape B=%d N=%d d=%d (d a multiple of 8)", B, N, d);
    if (!ok_dtype(act_dtype)) return set_error(VLG_ERR_DTYPE, "langfeat_arc_out: act_dtype %d", act_dtype);
    if (B == 0) return 0;
    if (!tri || !txt) return set_error(VLG_ERR_ARG, "langfeat_arc_out: null buffer");
    const size_t threads = (size_t)B * N * (d / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (act_dtype == VLG_F32)
        hipLaunchKernelGGL(langfeat_arc_out_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, tri, (const float*)aff, B, N, d, (float*)txt);
    else
        hipLaunchKernelGGL(langfeat_arc_out_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, tri, (const uint16_t*)aff, B, N, d, (uint16_t*)txt);
    return check_launch("langfeat_arc_out_kernel");
}

int vlg_langfeat_rowscale(const void* pre, int ld_in, const float* drop, int B, int N, int d, int ld_drop, int act_dtype, void* out, int ld_out, int width,
                          void* stream) {
    using namespace vlg;
    if (B < 0 || N < 1 || d < 8 || d % 8 || (drop && (ld_drop < d || ld_drop % 4)) || width < d || width % 8 || ld_in < d || ld_out < width || ld_in % 8 || ld_out % 8)
        return set_error(VLG_ERR_SHAPE, "langfeat_rowscale: bad shape B=%d N=%d d=%d ld_drop=%d ld_in=%d ld_out=%d width=%d (multiples of 8)", B, N, d, ld_drop,
                         ld_in, ld_out, width);
    if (!ok_dtype(act_dtype)) return set_error(VLG_ERR_DTYPE, "langfeat_rowscale: act_dtype %d", act_dtype);
    if (B == 0) return 0;
    if (!pre || !out) return set_error(VLG_ERR_ARG, "langfeat_rowscale: null buffer");
    const size_t threads = (size_t)B * N * (width / 8);
    const dim3 grid((unsigned)((threads + 255) / 256));
    if (act_dtype == VLG_F32)
        hipLaunchKernelGGL(langfeat_rowscale_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)pre, ld_in, drop, B, N, d, ld_drop, (float*)out,
                           ld_out, width);
    else
        hipLaunchKernelGGL(langfeat_rowscale_kernel<uint16_t>, grid, dim3(256), 0, (hipStream_t)stream, (const uint16_t*)pre, ld_in, drop, B, N, d, ld_drop,
                           (uint16_t*)out, ld_out, width);
    return check_launch("langfeat_rowscale_kernel");
}

}  // extern "C"
