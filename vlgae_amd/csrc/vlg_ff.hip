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

template <typename A>
__global__ __launch_bounds__(kFfThreads) void ff_act_kernel(const A* in, const A* __restrict__ res, const A* __restrict__ mask, float mask_scale, A* out,
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
    }
    store8(out + orow * H + c, v);
}

template <typename A>
__global__ __launch_bounds__(kFfThreads) void ff_act_bwd_kernel(const A* g, const A* __restrict__ act, const A* __restrict__ mask, float mask_scale, A* out,
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

int vlg_ff_act(const void* in, const void* residual, const void* mask, float mask_scale, void* out, long long M, int J, int H, int swap, int act_dtype,
               float slope, void* stream) {
    using namespace vlg;
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
                           mask_scale, (uint16_t*)out, rows, J, H, swap, slope);
    else
        hipLaunchKernelGGL(ff_act_kernel<float>, grid, dim3(kFfThreads), 0, s, (const float*)in, (const float*)residual, (const float*)mask, mask_scale,
                           (float*)out, rows, J, H, swap, slope);
    return check_launch("ff_act_kernel");
}

int vlg_ff_act_backward(const void* g, const void* act, const void* mask, float mask_scale, void* out, float* sum, long long M, int J, int H, int swap,
                        int accumulate, int act_dtype, float slope, void* stream) {
    using namespace vlg;
    if (M < 0 || J < 1 || (swap && J != 4)) return set_error(VLG_ERR_SHAPE, "ff_act_backward: M=%lld J=%d swap=%d", M, J, swap);
    if (int rc = ff_check("ff_act_backward", M * J, H, act_dtype)) return rc;
    if (M == 0) return 0;
    if (!g || !act || !out || (swap && g == out)) return set_error(VLG_ERR_ARG, "ff_act_backward: null buffer, or in-place with the permutation");
    if (!ff_aligned({g, act, mask, out, sum})) return set_error(VLG_ERR_ARG, "ff_act_backward: buffers must be 16-byte aligned");
    const dim3 grid = ff_grid((size_t)M * (H / 8));
    hipStream_t s = (hipStream_t)stream;
    if (act_dtype == VLG_BF16)
        hipLaunchKernelGGL(ff_act_bwd_kernel<uint16_t>, grid, dim3(kFfThreads), 0, s, (const uint16_t*)g, (const uint16_t*)act, (const uint16_t*)mask,
                           mask_scale, (uint16_t*)out, sum, (size_t)M, J, H, swap, accumulate, slope);
    else
        hipLaunchKernelGGL(ff_act_bwd_kernel<float>, grid, dim3(kFfThreads), 0, s, (const float*)g, (const float*)act, (const float*)mask, mask_scale,
                           (float*)out, sum, (size_t)M, J, H, swap, accumulate, slope);
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

}  // extern "C"
