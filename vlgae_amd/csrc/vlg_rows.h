// vlg_rows.h -- row helpers shared by the HBM-bound element-wise kernels (vlg_langfeat.hip, vlg_ff.hip): activations are stored as
// bf16 (uint16_t) or fp32 (float), arithmetic is fp32, one thread handles eight consecutive channels of a row (16 / 32 bytes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlg {

namespace {

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) {   // round to nearest even, like torch's cast
    const uint32_t u = __float_as_uint(v);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float ldf(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const uint16_t* p, size_t i) { return bf2f(p[i]); }

__device__ __forceinline__ void stf(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void stf(uint16_t* p, size_t i, float v) { p[i] = f2bf(v); }

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// eight consecutive channels of a row, as floats
__device__ __forceinline__ void load8(const uint16_t* p, float (&o)[8]) {
    const uint4 w = *reinterpret_cast<const uint4*>(p);
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(u[i] << 16); o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void store8(uint16_t* p, const float (&v)[8]) {
    uint4 w;
    w.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); w.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    w.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); w.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = w;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
// four consecutive channels (8 / 16 bytes)
__device__ __forceinline__ void load4(const uint16_t* p, float (&o)[4]) {
    const uint2 w = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(w.x << 16); o[1] = __uint_as_float(w.x & 0xffff0000u);
    o[2] = __uint_as_float(w.y << 16); o[3] = __uint_as_float(w.y & 0xffff0000u);
}
__device__ __forceinline__ void load4(const float* p, float (&o)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w;
}
__device__ __forceinline__ void store4(uint16_t* p, const float (&v)[4]) {
    uint2 w;
    w.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); w.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    *reinterpret_cast<uint2*>(p) = w;
}
__device__ __forceinline__ void store4(float* p, const float (&v)[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
// the value a tensor of storage type A holds after a store (bf16: rounded; fp32: itself)
template <typename A> __device__ __forceinline__ float stored(float v);
template <> __device__ __forceinline__ float stored<uint16_t>(float v) { return bf2f(f2bf(v)); }
template <> __device__ __forceinline__ float stored<float>(float v) { return v; }

}  // namespace

}  // namespace vlg
