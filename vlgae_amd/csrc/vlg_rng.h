// vlg_rng.h -- the counter-based dropout draw shared by the element-wise kernels (vlg_encoders.hip, vlg_ff.hip): Philox4x32-10
// (Salmon et al., SC'11) keyed by a DEVICE-resident (seed, step) pair, counter = element-group index.  Nothing is stored: the adjoint
// regenerates the bits of the forward pass, and a replayed HIP graph sees the step that vlg_rng_advance left in device memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vlg {

namespace {

__device__ __forceinline__ uint4 philox4x32_10(uint4 ctr, uint2 key) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, ctr.x), lo0 = 0xD2511F53u * ctr.x;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, ctr.z), lo1 = 0xCD9E8D57u * ctr.z;
        ctr = make_uint4(hi1 ^ ctr.y ^ key.x, lo1, hi0 ^ ctr.w ^ key.y, lo0);
        key.x += 0x9E3779B9u;
        key.y += 0xBB67AE85u;
    }
    return ctr;
}

// keep flags of eight consecutive elements (group index g): 16 random bits each, keep <=> bits >= thr, thr = round(p * 65536);
// m[k] = `scale` where kept, 0 where dropped.  `site`: which dropout layer of the step draws (independent streams off one state).
__device__ __forceinline__ void keep8(const uint64_t* __restrict__ rng, uint32_t site, uint64_t g, uint32_t thr, float scale, float (&m)[8]) {
    // The site goes into the key's second word on its own (times an odd constant), NOT into the seed's arithmetic: with `seed + site` the
    // stream of (seed s, site 2) was that of (seed s + 1, site 1) -- a trainer seeding its ranks with seed + rank would have drawn rank 0's
    // mid_ff mask and rank 1's text-encoder mask from the same bits.
    const uint64_t seed = rng[0], step = rng[1];
    const uint4 r = philox4x32_10(make_uint4((uint32_t)g, (uint32_t)(g >> 32), (uint32_t)step, (uint32_t)(step >> 32)),
                                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (site * 0x9E3779B9u)));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        m[2 * k] = (w[k] & 0xffffu) >= thr ? scale : 0.f;
        m[2 * k + 1] = (w[k] >> 16) >= thr ? scale : 0.f;
    }
}

inline uint32_t drop_threshold(float p) { return (uint32_t)(p * 65536.f + 0.5f); }          // p in steps of 2^-16
inline float drop_scale(float p) { return 1.f / (1.f - (float)drop_threshold(p) / 65536.f); }

}  // namespace

}  // namespace vlg
