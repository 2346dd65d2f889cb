// vlg_rel.hip -- the pairwise relation features of the visual encoder (gfx950), SURVEY.md section 8 f2.
//
//   VisBoxRelSimpleEncoder.forward, src/model/vis_encoder/box_rel.py:41-45:
//       _rel_inp = (inputs.unsqueeze(1) + inputs.unsqueeze(2)) / 2          [B,R,R,n_in]   (n_in = 4096: 5.1 GB at B = 256)
//       rel      = LeakyReLU(Linear(_rel_inp))                              [B,R*R,H]      (657 GFLOP at B = 256)
//   The Linear is linear:  W ((x_i + x_j)/2) + b = (y_i + y_j)/2 + b  with  y = W x  -- ONE [B R, n_in] x [n_in, H] library
//   GEMM (35x fewer flops) and the broadcast-add + activation below.  Neither the pairwise-mean tensor nor a second GEMM
//   operand ever exists.  What is left is HBM-bound byte work: write [B,R,R,H] once (forward), read its cotangent
//   (backward), coalesced 16-byte accesses, no matrix cores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_dp_core.h"   // F32In / BF16In

namespace vlg {

__device__ __forceinline__ float lrelu(float x, float slope) { return x > 0.f ? x : x * slope; }

__device__ __forceinline__ uint16_t f32_to_bf16(float f) {   // round to nearest even (finite inputs)
    const uint32_t u = __float_as_uint(f);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

template <typename In>
struct Vec4 {
    static __device__ __forceinline__ float4 ld(const typename In::T* p);
    static __device__ __forceinline__ void st(typename In::T* p, float4 v);
};
template <>
__device__ __forceinline__ float4 Vec4<F32In>::ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ void Vec4<F32In>::st(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
template <>
__device__ __forceinline__ float4 Vec4<BF16In>::ld(const uint16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}
template <>
__device__ __forceinline__ void Vec4<BF16In>::st(uint16_t* p, float4 v) {
    uint2 u;
    u.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    u.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}

// One block = one (b, i): row y[b,i,:] stays in registers, the block sweeps j.  Thread = 4 consecutive channels of one j.
template <typename In>
__global__ __launch_bounds__(256) void box_rel_fwd_kernel(const typename In::T* __restrict__ y, const float* __restrict__ bias,
                                                          int R, int H, float slope, typename In::T* __restrict__ out) {
    const int b = blockIdx.y, i = blockIdx.x, tpr = H >> 2;            // threads per row
    const int c4 = (threadIdx.x % tpr) * 4, jl = threadIdx.x / tpr, jstep = 256 / tpr;
    const typename In::T* yb = y + (size_t)b * R * H;
    const float4 yi = Vec4<In>::ld(yb + (size_t)i * H + c4);
    const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    typename In::T* ob = out + ((size_t)b * R + i) * R * H;
    for (int j = jl; j < R; j += jstep) {
        const float4 yj = Vec4<In>::ld(yb + (size_t)j * H + c4);
        float4 o;
        o.x = lrelu(0.5f * (yi.x + yj.x) + bb.x, slope);
        o.y = lrelu(0.5f * (yi.y + yj.y) + bb.y, slope);
        o.z = lrelu(0.5f * (yi.z + yj.z) + bb.z, slope);
        o.w = lrelu(0.5f * (yi.w + yj.w) + bb.w, slope);
        Vec4<In>::st(ob + (size_t)j * H + c4, o);
    }
}

// Adjoint.  pre[b,i,j,:] = (y_i + y_j)/2 + bias is symmetric in (i, j), so with s = LeakyReLU'(pre):
//   g_y[b,i,:] = 1/2 sum_j s[i,j] (g[b,i,j,:] + g[b,j,i,:])        g_bias = sum_{b,i,j} s[i,j] g[b,i,j,:]
// One block = one (b, i); both cotangent rows it needs are contiguous 16-byte-aligned rows.  The bias partial of the
// block (direct terms only) goes to a [B R, H] scratch that a second launch column-sums in a fixed order.
template <typename In>
__global__ __launch_bounds__(256) void box_rel_bwd_kernel(const typename In::T* __restrict__ y, const float* __restrict__ bias,
                                                          const typename In::T* __restrict__ g, int R, int H, float slope,
                                                          float* __restrict__ g_y, float* __restrict__ bias_part) {
    __shared__ float4 red[256];
    __shared__ float4 redb[256];
    const int b = blockIdx.y, i = blockIdx.x, tpr = H >> 2;
    const int c4 = (threadIdx.x % tpr) * 4, jl = threadIdx.x / tpr, jstep = 256 / tpr;
    const typename In::T* yb = y + (size_t)b * R * H;
    const float4 yi = Vec4<In>::ld(yb + (size_t)i * H + c4);
    const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const typename In::T* gb = g + (size_t)b * R * R * H;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f), accb = acc;
    for (int j = jl; j < R; j += jstep) {
        const float4 yj = Vec4<In>::ld(yb + (size_t)j * H + c4);
        const float4 gij = Vec4<In>::ld(gb + ((size_t)i * R + j) * H + c4);
        const float4 gji = Vec4<In>::ld(gb + ((size_t)j * R + i) * H + c4);
        const float sx = 0.5f * (yi.x + yj.x) + bb.x > 0.f ? 1.f : slope, sy = 0.5f * (yi.y + yj.y) + bb.y > 0.f ? 1.f : slope;
        const float sz = 0.5f * (yi.z + yj.z) + bb.z > 0.f ? 1.f : slope, sw = 0.5f * (yi.w + yj.w) + bb.w > 0.f ? 1.f : slope;
        acc.x += sx * (gij.x + gji.x); acc.y += sy * (gij.y + gji.y); acc.z += sz * (gij.z + gji.z); acc.w += sw * (gij.w + gji.w);
        accb.x += sx * gij.x; accb.y += sy * gij.y; accb.z += sz * gij.z; accb.w += sw * gij.w;
    }
    red[threadIdx.x] = acc;
    redb[threadIdx.x] = accb;
    __syncthreads();
    if (jl == 0) {   // fixed-order sum over the block's j-slices
        for (int k = 1; k < jstep; ++k) {
            const float4 a = red[threadIdx.x + k * tpr], ab = redb[threadIdx.x + k * tpr];
            acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
            accb.x += ab.x; accb.y += ab.y; accb.z += ab.z; accb.w += ab.w;
        }
        const size_t row = (size_t)b * R + i;
        *reinterpret_cast<float4*>(g_y + row * H + c4) = make_float4(0.5f * acc.x, 0.5f * acc.y, 0.5f * acc.z, 0.5f * acc.w);
        if (bias_part) *reinterpret_cast<float4*>(bias_part + row * H + c4) = accb;
    }
}

// out[h] = sum_r src[r][h], fixed order: one block = 64 columns x 16 row groups
__global__ __launch_bounds__(1024) void colsum_kernel(const float* __restrict__ src, int rows, int H, float* __restrict__ out) {
    __shared__ float part[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < H)
        for (int r = rg; r < rows; r += 16) acc += src[(size_t)r * H + col];
    part[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && col < H) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
        out[col] = t;
    }
}

static int rel_check(const char* what, int B, int R, int H, int dtype) {
    if (B < 1 || R < 1 || H < 4) return set_error(VLG_ERR_SHAPE, "%s: bad shape B=%d R=%d H=%d", what, B, R, H);
    if (H % 4 != 0 || H > 1024 || 256 % (H / 4) != 0)
        return set_error(VLG_ERR_SHAPE, "%s: H=%d must be a multiple of 4 with H/4 dividing 256 (the model's 256)", what, H);
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "%s: B=%d exceeds grid.y", what, B);
    if (dtype != VLG_F32 && dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "%s: dtype %d", what, dtype);
    return 0;
}

}  // namespace vlg

extern "C" {

int vlg_box_rel_pairwise(const void* y, const float* bias, int B, int R, int H, int dtype, float slope, void* out, void* stream) {
    using namespace vlg;
    if (int rc = rel_check("box_rel_pairwise", B, R, H, dtype)) return rc;
    if (!y || !out) return set_error(VLG_ERR_ARG, "box_rel_pairwise: null buffer");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VLG_F32)
        hipLaunchKernelGGL(box_rel_fwd_kernel<F32In>, dim3(R, B), dim3(256), 0, s, (const float*)y, bias, R, H, slope, (float*)out);
    else
        hipLaunchKernelGGL(box_rel_fwd_kernel<BF16In>, dim3(R, B), dim3(256), 0, s, (const uint16_t*)y, bias, R, H, slope, (uint16_t*)out);
    return check_launch("box_rel_fwd_kernel");
}

size_t vlg_box_rel_pairwise_backward_workspace(int B, int R, int H) {
    if (B < 1 || R < 1 || H < 1) return 0;
    return sizeof(float) * (size_t)B * R * H;
}

int vlg_box_rel_pairwise_backward(const void* y, const float* bias, const void* grad_out, int B, int R, int H, int dtype, float slope,
                                  void* ws, size_t ws_bytes, float* grad_y, float* grad_bias, void* stream) {
    using namespace vlg;
    if (int rc = rel_check("box_rel_pairwise_backward", B, R, H, dtype)) return rc;
    if (!y || !grad_out || !grad_y) return set_error(VLG_ERR_ARG, "box_rel_pairwise_backward: null buffer");
    if (grad_bias && (!ws || ws_bytes < vlg_box_rel_pairwise_backward_workspace(B, R, H)))
        return set_error(VLG_ERR_WORKSPACE, "box_rel_pairwise_backward: workspace %zu bytes < %zu", ws_bytes,
                         vlg_box_rel_pairwise_backward_workspace(B, R, H));
    hipStream_t s = (hipStream_t)stream;
    float* part = grad_bias ? (float*)ws : nullptr;
    if (dtype == VLG_F32)
        hipLaunchKernelGGL(box_rel_bwd_kernel<F32In>, dim3(R, B), dim3(256), 0, s, (const float*)y, bias, (const float*)grad_out, R, H,
                           slope, grad_y, part);
    else
        hipLaunchKernelGGL(box_rel_bwd_kernel<BF16In>, dim3(R, B), dim3(256), 0, s, (const uint16_t*)y, bias,
                           (const uint16_t*)grad_out, R, H, slope, grad_y, part);
    if (int rc = check_launch("box_rel_bwd_kernel")) return rc;
    if (grad_bias) {
        hipLaunchKernelGGL(colsum_kernel, dim3((H + 63) / 64), dim3(1024), 0, s, part, B * R, H, grad_bias);
        return check_launch("colsum_kernel");
    }
    return 0;
}

}  // extern "C"
