// vlg_ff.hip -- the element-wise passes BETWEEN the library GEMMs of the parser's feed-forwards (vlgae_amd/parser_ff.py):
// `MLP` (src/model/nn/common.py:23-51: Linear -> LeakyReLU -> SharedDropout) and `DMVSkipConnectEncoder`
// (src/model/nn/dmv_spec.py:38-54: bottleneck + skip connection -> LeakyReLU -> Linear -> LeakyReLU, twice, -> nn.Dropout).
// As torch runs them these are 2-4 launches per stage (broadcast add, LeakyReLU, mask multiply, permuted copy, group sum),
// each a full pass over a [4 B L, H] activation (21 MB at B = 256, L = 40, H = 256 in bf16); here every stage is ONE pass:
//
//   ff_mlp_act_kernel       nn/common.py:47-51    X <- LeakyReLU(X + context term of the sentence) * SharedDropout mask, in place
//   ff_act_kernel           nn/dmv_spec.py:41-52  out[m,j'] = LeakyReLU(in[m,j] + x[m]) * mask[m,j']; j' = j, or (dir,val) <- (val,dir)
//                                                 (the torch.stack(dim=-3) of :47 as a store permutation instead of a copy)
//   ff_act_bwd_kernel                             out[m,j'] = LeakyReLU'(act[m,j]) * (g[m,j] * mask[m,j]); sum[m] (+)= sum_j of it
//                                                 (the skip connection's cotangent), same optional permutation
//   ff_mlp_act_bwd_kernel                         gpre = LeakyReLU'(X) * mask * (gX + T)
//   ff_colmean_kernel       ldndmv.py:226         the sentence's context vector: mean over the positions, cast included
// Storage type A (bf16 / fp32) as in vlg_langfeat.hip; arithmetic in fp32, ONE rounding per stored element (torch rounds after
// every launch of the chain it replaces).  LeakyReLU' is taken from the sign of the activation's stored OUTPUT (the mask multiply
// cannot flip it; where the mask is 0 the cotangent is 0 as well).  One thread = eight channels of a row; H a multiple of 8.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <initializer_list>

#include "vlg_common.h"
#include "vlg_rng.h"
#include "vlg_rows.h"

namespace vlg {

namespace {

constexpr int kFfThreads = 256;

// (dir, val) <-> (val, dir) of a row index j in [0, 4)
__device__ __forceinline__ int swap2(int j) { return ((j & 1) << 1) | (j >> 1); }

template <typename A>
__global__ __launch_bounds__(kFfThreads) void ff_mlp_act_kernel(A* X, const A* __restrict__ cterm, const float* __restrict__ drop_head,
                                                                const float* __restrict__ drop_small, int M0, int Ms, int L, int H, float slope) {
    const int hv = H >> 3;
    const size_t i = (size_t)blockIdx.x * kFfThreads + threadIdx.x;
    if (i >= (size_t)(M0 + Ms) * hv) return;
    const int row = (int)(i / hv), c = (int)(i - (size_t)row * hv) * 8;
    A* p = X + (size_t)row * H + c;
    float v[8];
    load8(p, v);
    if (row < M0) {   // a parent row: + W_ctx mean(x) + bias of its sentence; the sentence's SharedDropout mask
        const int b = row / L;
        float t[8];
        load8(cterm + (size_t)b * H + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = leaky(v[k] + t[k], slope);
        if (drop_head) {
            const float4 m0 = *reinterpret_cast<const float4*>(drop_head + (size_t)b * H + c);
            const float4 m1 = *reinterpret_cast<const float4*>(drop_head + (size_t)b * H + c + 4);
            v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w; v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
        }
    } else {          // a token / root / decision row (2-D input of its MLP: one mask value per row, nn/dropout.py:52-53)
        const float m = drop_small ? drop_small[row - M0] : 1.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = leaky(v[k], slope) * m;
    }
    store8(p, v);
}

// mask: an explicit keep-mask (times mask_scale), or -- rng != null -- the counter-based draw over the OUTPUT element index (mask_scale =
// 1 / (1 - p), thr = round(p 2^16)): nn.Dropout without a mask tensor (21 MB per step at B = 256, L = 40 for mid_ff's, nn/dmv_spec.py:52)
template <typename A>
__global__ __launch_bounds__(kFfThreads) void ff_act_kernel(const A* in, const A* __restrict__ res, const A* __restrict__ mask, float mask_scale,
                                                            const uint64_t* __restrict__ rng, uint32_t site, uint32_t thr, A* out,
                                                            size_t rows, int J, int H, int swap, float slope) {
    const int hv = H >> 3;
    const size_t i = (size_t)blockIdx.x * kFfThreads + threadIdx.x;
    if (i >= rows * hv) return;
    const size_t row = i / hv;
    const int c = (int)(i - row * hv) * 8;
    const size_t m = row / J;
    const int j = (int)(row - m * J);
    const size_t orow = swap ? m * 4 + swap2(j) : row;
    float v[8];
    load8(in + row * H + c, v);
    if (res) {
        float t[8];
        load8(res + m * H + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = leaky(v[k], slope);
    if (mask) {
        float t[8];
        load8(mask + orow * H + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= t[k] * mask_scale;
    } else if (rng) {
        float t[8];
        keep8(rng, site, orow * hv + (c >> 3), thr, mask_scale, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= t[k];
    }
    store8(out + orow * H + c, v);
}

template <typename A>
__global__ __launch_bounds__(kFfThreads) void ff_act_bwd_kernel(const A* g, const A* __restrict__ act, const A* __restrict__ mask, float mask_scale,
                                                                const uint64_t* __restrict__ rng, uint32_t site, uint32_t thr, A* out,
                                                                float* sum, size_t M, int J, int H, int swap, int accumulate, float slope) {
    const int hv = H >> 3;
    const size_t i = (size_t)blockIdx.x * kFfThreads + threadIdx.x;
    if (i >= M * hv) return;
    const size_t m = i / hv;
    const int c = (int)(i - m * hv) * 8;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (sum && accumulate) load8(sum + m * H + c, s);
    for (int j = 0; j < J; ++j) {
        const size_t row = m * J + j, orow = swap ? m * 4 + swap2(j) : row;
        float gv[8], av[8];
        load8(g + row * H + c, gv);
        load8(act + row * H + c, av);
        if (mask) {
            float t[8];
            load8(mask + row * H + c, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) gv[k] *= t[k] * mask_scale;
        } else if (rng) {       // the forward pass's bits again (its mask index is the row it WROTE; no permutation with a mask: checked on the host)
            float t[8];
            keep8(rng, site, row * hv + (c >> 3), thr, mask_scale, t);
#pragma unroll
            for (int k = 0; k < 8; ++k) gv[k] *= t[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            gv[k] = av[k] > 0.f ? gv[k] : gv[k] * slope;
            s[k] += stored<A>(gv[k]);      // the sum of what the next GEMM reads
        }
        store8(out + orow * H + c, gv);
    }
    if (sum) store8(sum + m * H + c, s);
}

template <typename A>
__global__ __launch_bounds__(kFfThreads) void ff_mlp_act_bwd_kernel(const float* __restrict__ gX, const A* __restrict__ T, const A* __restrict__ X,
                                                                    const float* __restrict__ drop_head, const float* __restrict__ drop_small,
                                                                    A* __restrict__ gpre, int M0, int Ms, int L, int H, float slope) {
    const int hv = H >> 3;
    const size_t i = (size_t)blockIdx.x * kFfThreads + threadIdx.x;
    if (i >= (size_t)(M0 + Ms) * hv) return;
    const int row = (int)(i / hv), c = (int)(i - (size_t)row * hv) * 8;
    const size_t o = (size_t)row * H + c;
    float v[8], xv[8];
    load8(gX + o, v);
    load8(X + o, xv);
    if (T) {
        float t[8];
        load8(T + o, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    if (row < M0) {
        if (drop_head) {
            const size_t b = (size_t)(row / L) * H + c;
            const float4 m0 = *reinterpret_cast<const float4*>(drop_head + b), m1 = *reinterpret_cast<const float4*>(drop_head + b + 4);
            v[0] *= m0.x; v[1] *= m0.y; v[2] *= m0.z; v[3] *= m0.w; v[4] *= m1.x; v[5] *= m1.y; v[6] *= m1.z; v[7] *= m1.w;
        }
    } else if (drop_small) {
        const float m = drop_small[row - M0];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= m;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = xv[k] > 0.f ? v[k] : v[k] * slope;
    store8(gpre + o, v);
}

// context_mode 'mean' (src/model/ldndmv.py:226): out[b,:] = mean_l x[b,l,:] over ALL L positions, any (float32 | bf16) -> any: torch
// does this as a cast launch + a reduction launch.  One thread per (sentence, channel), rows read coalesced across the threads.
template <typename T, typename A>
__global__ __launch_bounds__(256) void ff_colmean_kernel(const T* __restrict__ x, int L, int h, A* __restrict__ out) {
    const int b = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    if (c >= h) return;
    const T* src = x + (size_t)b * L * h + c;
    float acc = 0.f;
#pragma unroll 4
    for (int l = 0; l < L; ++l) acc += stored<A>(ldf(src, (size_t)l * h));   // (the values a cast to the activations' dtype would have produced)
    stf(out, (size_t)b * h + c, acc / (float)L);
}

// The root rule (ldndmv.py:205): logits[c] = sum_{dv,e} r1[dv,e] r2[c,dv,e] (root_scorer(h_root, h_child).sum over (dir, val)),
// root_rule = log_softmax over the T tokens.  r1 / r2 are blocks of the small projection product S [4 (T + 3), ld] (rows (c, dv) of the
// tokens, then the root's four (dir, val) rows): r2 = S[4c + dv, col2..col2+r), r1 = S[4T + dv, col1..col1+r).  One workgroup.
template <typename A>
__global__ __launch_bounds__(256) void ff_root_rule_kernel(const A* __restrict__ S, int ld, int T, int r, int col1, int col2, float* __restrict__ root_rule) {
    extern __shared__ float sh[];          // T logits + 256 reduction slots
    float* logit = sh;
    float* red = sh + T;
    const int tid = threadIdx.x;
    for (int c = tid; c < T; c += 256) {
        float acc = 0.f;
        for (int dv = 0; dv < 4; ++dv)
            for (int e = 0; e < r; ++e) acc = fmaf(ldf(S, (size_t)(4 * T + dv) * ld + col1 + e), ldf(S, (size_t)(4 * c + dv) * ld + col2 + e), acc);
        logit[c] = acc;
    }
    __syncthreads();
    float m = -3.4e38f;
    for (int c = tid; c < T; c += 256) m = fmaxf(m, logit[c]);
    red[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
        __syncthreads();
    }
    m = red[0];
    __syncthreads();
    float z = 0.f;
    for (int c = tid; c < T; c += 256) z += expf(logit[c] - m);
    red[tid] = z;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float lz = m + logf(red[0]);
    for (int c = tid; c < T; c += 256) root_rule[c] = logit[c] - lz;
}

// Its adjoint, and the whole cotangent of S assembled in ONE pass: gS [4 (T + 3), 4r] (column blocks x2 | r2 | r1 | y2) =
//   rows 4c+dv, block 0: g_x2 (the scorer's cotangent of attach.project2)       block 1: dlogit[c] r1[dv,:]
//   rows 4T+dv, block 2: sum_c dlogit[c] r2[c,dv,:]                             rows 4T+4.., block 3: g_y2 (8 rows)        zeros elsewhere
// with dlogit = g_root - softmax * sum(g_root).  One workgroup per 64 rows of gS; every workgroup recomputes the T-term sum of g_root.
template <typename A>
__global__ __launch_bounds__(256) void ff_root_rule_bwd_kernel(const A* __restrict__ S, int ld, int T, int r, const float* __restrict__ root_rule,
                                                               const float* __restrict__ g_root, const A* __restrict__ g_x2, int ld_x2,
                                                               const A* __restrict__ g_y2, int ld_y2, A* __restrict__ gS) {
    __shared__ float red[256];
    extern __shared__ float dl[];          // dlogit[c] = g_root[c] - softmax[c] sum(g_root): once per workgroup (a serial chain of dependent
                                           // global loads per output element made this kernel 17.8 us for 12 K outputs)
    const int tid = threadIdx.x, rows = 4 * (T + 3), W = 4 * r;
    float t = 0.f;
    for (int c = tid; c < T; c += 256) t += g_root[c];
    red[tid] = t;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] += red[tid + s];
        __syncthreads();
    }
    const float gsum = red[0];
    for (int c = tid; c < T; c += 256) dl[c] = g_root[c] - expf(root_rule[c]) * gsum;
    __syncthreads();
    for (int i = blockIdx.x * 64 * W + tid; i < min(rows, (int)(blockIdx.x + 1) * 64) * W; i += 256) {
        const int row = i / W, col = i - row * W, blk = col / r, e = col - blk * r;
        float v = 0.f;
        if (row < 4 * T) {
            const int c = row >> 2, dv = row & 3;
            if (blk == 0) v = g_x2 ? ldf(g_x2, (size_t)row * ld_x2 + e) : 0.f;
            else if (blk == 1) v = dl[c] * ldf(S, (size_t)(4 * T + dv) * ld + 2 * r + e);
        } else if (row < 4 * T + 4) {
            if (blk == 2) {
                const int dv = row - 4 * T;
#pragma unroll 8
                for (int c = 0; c < T; ++c) v = fmaf(dl[c], ldf(S, (size_t)(4 * c + dv) * ld + r + e), v);   // (independent loads: eight in flight)
            }
        } else if (blk == 3) {
            v = g_y2 ? ldf(g_y2, (size_t)(row - 4 * T - 4) * ld_y2 + e) : 0.f;
        }
        stf(gS, (size_t)i, v);
    }
}

int ff_check(const char* what, long long rows, int H, int act_dtype) {
    if (rows < 0 || H < 8 || H % 8) return set_error(VLG_ERR_SHAPE, "%s: rows=%lld H=%d (H must be a positive multiple of 8)", what, rows, H);
    if (act_dtype != VLG_F32 && act_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "%s: act_dtype %d", what, act_dtype);
    if ((unsigned long long)rows * (unsigned)(H / 8) > 0x7fffffffull * kFfThreads) return set_error(VLG_ERR_SHAPE, "%s: rows=%lld too large", what, rows);
    return 0;
}

inline dim3 ff_grid(size_t threads) { return dim3((unsigned)((threads + kFfThreads - 1) / kFfThreads)); }

// rows are read and written 16 bytes at a time
inline bool ff_aligned(std::initializer_list<const void*> ps) {
    for (const void* p : ps)
        if (reinterpret_cast<uintptr_t>(p) & 15) return false;
    return true;
}

}  // namespace

}  // namespace vlg

extern "C" {

int vlg_ff_mlp_act(void* x, const void* cterm, const float* drop_head, const float* drop_small, int B, int L, int Ms, int H, int act_dtype,
                   float slope, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || Ms < 0) return set_error(VLG_ERR_SHAPE, "ff_mlp_act: B=%d L=%d Ms=%d", B, L, Ms);
    const long long M0 = (long long)B * L;
    if (int rc = ff_check("ff_mlp_act", M0 + Ms, H, act_dtype)) return rc;
    if (M0 + Ms == 0) return 0;
    if (!x || (M0 > 0 && !cterm)) return set_error(VLG_ERR_ARG, "ff_mlp_act: null buffer");
    if (!ff_aligned({x, cterm, drop_head})) return set_error(VLG_ERR_ARG, "ff_mlp_act: x, cterm and drop_head must be 16-byte aligned");
    const dim3 grid = ff_grid((size_t)(M0 + Ms) * (H / 8));
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16)
        hipLaunchKernelGGL(ff_mlp_act_kernel<uint16_t>, grid, dim3(kFfThreads), 0, s, (uint16_t*)x, (const uint16_t*)cterm, drop_head, drop_small,
                           (int)M0, Ms, L, H, slope);
    else
        hipLaunchKernelGGL(ff_mlp_act_kernel<float>, grid, dim3(kFfThreads), 0, s, (float*)x, (const float*)cterm, drop_head, drop_small, (int)M0,
                           Ms, L, H, slope);
    return check_launch("ff_mlp_act_kernel");
}

int vlg_ff_act(const void* in, const void* residual, const void* mask, float mask_scale, const uint64_t* rng, unsigned site, float p, void* out,
               long long M, int J, int H, int swap, int act_dtype, float slope, void* stream) {
    using namespace vlg;
    if (rng && (mask || swap || !(p >= 0.f && p < 1.f))) return set_error(VLG_ERR_ARG, "ff_act: the counter-based draw excludes an explicit mask and the permutation; p=%f", (double)p);
    const uint32_t thr = rng ? drop_threshold(p) : 0;
    if (rng) mask_scale = drop_scale(p);
    if (M < 0 || J < 1 || (swap && J != 4)) return set_error(VLG_ERR_SHAPE, "ff_act: M=%lld J=%d swap=%d (the permutation is of J = 4 = (val, dir))", M, J, swap);
    if (int rc = ff_check("ff_act", M * J, H, act_dtype)) return rc;
    if (M == 0) return 0;
    if (!in || !out || (swap && in == out)) return set_error(VLG_ERR_ARG, "ff_act: null buffer, or in-place with the permutation");
    if (!ff_aligned({in, residual, mask, out})) return set_error(VLG_ERR_ARG, "ff_act: buffers must be 16-byte aligned");
    const size_t rows = (size_t)M * J;
    const dim3 grid = ff_grid(rows * (H / 8));
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16)
        hipLaunchKernelGGL(ff_act_kernel<uint16_t>, grid, dim3(kFfThreads), 0, s, (const uint16_t*)in, (const uint16_t*)residual, (const uint16_t*)mask,
                           mask_scale, rng, site, thr, (uint16_t*)out, rows, J, H, swap, slope);
    else
        hipLaunchKernelGGL(ff_act_kernel<float>, grid, dim3(kFfThreads), 0, s, (const float*)in, (const float*)residual, (const float*)mask, mask_scale,
                           rng, site, thr, (float*)out, rows, J, H, swap, slope);
    return check_launch("ff_act_kernel");
}

int vlg_ff_act_backward(const void* g, const void* act, const void* mask, float mask_scale, const uint64_t* rng, unsigned site, float p, void* out,
                        float* sum, long long M, int J, int H, int swap, int accumulate, int act_dtype, float slope, void* stream) {
    using namespace vlg;
    if (rng && (mask || swap || !(p >= 0.f && p < 1.f))) return set_error(VLG_ERR_ARG, "ff_act_backward: the counter-based draw excludes an explicit mask and the permutation; p=%f", (double)p);
    const uint32_t thr = rng ? drop_threshold(p) : 0;
    if (rng) mask_scale = drop_scale(p);
    if (M < 0 || J < 1 || (swap && J != 4)) return set_error(VLG_ERR_SHAPE, "ff_act_backward: M=%lld J=%d swap=%d", M, J, swap);
    if (int rc = ff_check("ff_act_backward", M * J, H, act_dtype)) return rc;
    if (M == 0) return 0;
    if (!g || !act || !out || (swap && g == out)) return set_error(VLG_ERR_ARG, "ff_act_backward: null buffer, or in-place with the permutation");
    if (!ff_aligned({g, act, mask, out, sum})) return set_error(VLG_ERR_ARG, "ff_act_backward: buffers must be 16-byte aligned");
    const dim3 grid = ff_grid((size_t)M * (H / 8));
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16)
        hipLaunchKernelGGL(ff_act_bwd_kernel<uint16_t>, grid, dim3(kFfThreads), 0, s, (const uint16_t*)g, (const uint16_t*)act, (const uint16_t*)mask,
                           mask_scale, rng, site, thr, (uint16_t*)out, sum, (size_t)M, J, H, swap, accumulate, slope);
    else
        hipLaunchKernelGGL(ff_act_bwd_kernel<float>, grid, dim3(kFfThreads), 0, s, (const float*)g, (const float*)act, (const float*)mask, mask_scale,
                           rng, site, thr, (float*)out, sum, (size_t)M, J, H, swap, accumulate, slope);
    return check_launch("ff_act_bwd_kernel");
}

int vlg_ff_mlp_act_backward(const float* gx, const void* t, const void* x, const float* drop_head, const float* drop_small, void* gpre, int B, int L,
                            int Ms, int H, int act_dtype, float slope, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || Ms < 0) return set_error(VLG_ERR_SHAPE, "ff_mlp_act_backward: B=%d L=%d Ms=%d", B, L, Ms);
    const long long M0 = (long long)B * L;
    if (int rc = ff_check("ff_mlp_act_backward", M0 + Ms, H, act_dtype)) return rc;
    if (M0 + Ms == 0) return 0;
    if (!gx || !x || !gpre) return set_error(VLG_ERR_ARG, "ff_mlp_act_backward: null buffer");
    if (!ff_aligned({gx, t, x, drop_head, gpre})) return set_error(VLG_ERR_ARG, "ff_mlp_act_backward: buffers must be 16-byte aligned");
    const dim3 grid = ff_grid((size_t)(M0 + Ms) * (H / 8));
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16)
        hipLaunchKernelGGL(ff_mlp_act_bwd_kernel<uint16_t>, grid, dim3(kFfThreads), 0, s, gx, (const uint16_t*)t, (const uint16_t*)x, drop_head,
                           drop_small, (uint16_t*)gpre, (int)M0, Ms, L, H, slope);
    else
        hipLaunchKernelGGL(ff_mlp_act_bwd_kernel<float>, grid, dim3(kFfThreads), 0, s, gx, (const float*)t, (const float*)x, drop_head, drop_small,
                           (float*)gpre, (int)M0, Ms, L, H, slope);
    return check_launch("ff_mlp_act_bwd_kernel");
}

int vlg_ff_context_mean(const void* x, int in_dtype, int B, int L, int h, void* out, int out_dtype, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || h < 1 || B > 0x7fffffff / 1) return set_error(VLG_ERR_SHAPE, "ff_context_mean: B=%d L=%d h=%d", B, L, h);
    if ((in_dtype != VLG_F32 && in_dtype != VLG_BF16) || (out_dtype != VLG_F32 && out_dtype != VLG_BF16))
        return set_error(VLG_ERR_DTYPE, "ff_context_mean: dtypes %d -> %d", in_dtype, out_dtype);
    if (B == 0) return 0;
    if (!x || !out) return set_error(VLG_ERR_ARG, "ff_context_mean: null buffer");
    const dim3 grid(B, (h + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == VLG_F32 && out_dtype == VLG_F32) hipLaunchKernelGGL((ff_colmean_kernel<float, float>), grid, dim3(256), 0, s, (const float*)x, L, h, (float*)out);
    else if (in_dtype == VLG_F32) hipLaunchKernelGGL((ff_colmean_kernel<float, uint16_t>), grid, dim3(256), 0, s, (const float*)x, L, h, (uint16_t*)out);
    else if (out_dtype == VLG_F32) hipLaunchKernelGGL((ff_colmean_kernel<uint16_t, float>), grid, dim3(256), 0, s, (const uint16_t*)x, L, h, (float*)out);
    else hipLaunchKernelGGL((ff_colmean_kernel<uint16_t, uint16_t>), grid, dim3(256), 0, s, (const uint16_t*)x, L, h, (uint16_t*)out);
    return check_launch("ff_colmean_kernel");
}

int vlg_ff_root_rule(const void* small, int ld, int T, int r, int act_dtype, float* root_rule, void* stream) {
    using namespace vlg;
    if (T < 1 || r < 1 || ld < 4 * r || T > 8192) return set_error(VLG_ERR_SHAPE, "ff_root_rule: T=%d r=%d ld=%d", T, r, ld);
    if (act_dtype != VLG_F32 && act_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "ff_root_rule: act_dtype %d", act_dtype);
    if (!small || !root_rule) return set_error(VLG_ERR_ARG, "ff_root_rule: null buffer");
    const size_t lds = sizeof(float) * ((size_t)T + 256);
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16) hipLaunchKernelGGL(ff_root_rule_kernel<uint16_t>, dim3(1), dim3(256), lds, s, (const uint16_t*)small, ld, T, r, 2 * r, r, root_rule);
    else hipLaunchKernelGGL(ff_root_rule_kernel<float>, dim3(1), dim3(256), lds, s, (const float*)small, ld, T, r, 2 * r, r, root_rule);
    return check_launch("ff_root_rule_kernel");
}

int vlg_ff_root_rule_backward(const void* small, int ld, int T, int r, int act_dtype, const float* root_rule, const float* g_root, const void* g_x2,
                              int ld_x2, const void* g_y2, int ld_y2, void* g_small, void* stream) {
    using namespace vlg;
    if (T < 1 || r < 1 || ld < 4 * r || T > 8192) return set_error(VLG_ERR_SHAPE, "ff_root_rule_backward: T=%d r=%d ld=%d", T, r, ld);
    if (act_dtype != VLG_F32 && act_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "ff_root_rule_backward: act_dtype %d", act_dtype);
    if (!small || !root_rule || !g_root || !g_small) return set_error(VLG_ERR_ARG, "ff_root_rule_backward: null buffer");
    if ((g_x2 && ld_x2 < r) || (g_y2 && ld_y2 < r)) return set_error(VLG_ERR_SHAPE, "ff_root_rule_backward: ld_x2=%d ld_y2=%d below r=%d", ld_x2, ld_y2, r);
    const dim3 grid((4 * (T + 3) + 63) / 64);
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16)
        hipLaunchKernelGGL(ff_root_rule_bwd_kernel<uint16_t>, grid, dim3(256), sizeof(float) * (size_t)T, s, (const uint16_t*)small, ld, T, r, root_rule, g_root, (const uint16_t*)g_x2, ld_x2,
                           (const uint16_t*)g_y2, ld_y2, (uint16_t*)g_small);
    else
        hipLaunchKernelGGL(ff_root_rule_bwd_kernel<float>, grid, dim3(256), sizeof(float) * (size_t)T, s, (const float*)small, ld, T, r, root_rule, g_root, (const float*)g_x2, ld_x2,
                           (const float*)g_y2, ld_y2, (float*)g_small);
    return check_launch("ff_root_rule_bwd_kernel");
}

}  // extern "C"
