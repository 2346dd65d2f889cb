// vlg_arc.hip -- the arc encoder's trilinear contraction on the matrix cores (gfx950) and its C-ABI entry points.
//
//   lang_feat word+maxdep, src/model/joint.py:281-284:   arc[b,c,h] = sum_{x,y} child[b,c,x] * w1[x,h,y] * parent[b,c,y]
//   (SURVEY.md section 8 f2).  torch evaluates the einsum pairwise and materialises [B,C,H,Y] (688 MB fp32 at B = 256,
//   C = 41, 128^3 weights) on the way; here the inner product over y is an MFMA tile that is scaled by child[m,x] and
//   added to the output accumulators without ever leaving registers.
//
//   tri_kernel      out[m,h] = sum_x c[m,x] * ( sum_y w[x,h,y] * p[m,y] )          (m = flattened batch x position)
//                   For each x the bracket is a [rows m] x [cols h] MFMA tile over K = y, contiguous on both operands
//                   (p rows and w[x,h,:] rows), so fragments are 16-byte reads straight from global memory.
//                   The SAME kernel gives both input gradients with a permuted copy of the weights:
//                     d_child [m,x] = sum_y p[m,y] * ( sum_h wA[y,x,h] * g[m,h] ),   wA[y,x,h] = w[x,h,y]
//                     d_parent[m,y] = sum_x c[m,x] * ( sum_h wB[x,y,h] * g[m,h] ),   wB[x,y,h] = w[x,h,y]
//   tri_dw_kernel   d_w[x,h,y] = sum_m c[m,x] g[m,h] p[m,y]: K = m, so the three operands are read from transposed copies
//                   ([dim][M], m contiguous); the A operand c[m,x] * g[m,h] is formed in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_mfma.h"

namespace vlg {

constexpr int kTriWaves = 4;  // waves per block; wave w takes h-tiles w, w + 4, ...
constexpr int kTriHPW = 2;    // h-tiles per wave: H <= 128

template <bool F32IN>
__device__ __forceinline__ float tri_ld(const typename MfmaCfg<F32IN>::T* p, size_t i) {
    if constexpr (F32IN) return p[i];
    else return __uint_as_float((uint32_t)p[i] << 16);
}

// out[m,h] (+)= sum_{x in this block's range} c[m,x] * sum_y w[x,h,y] * p[m,y];  Y == KCH * KW, H % 16 == 0, H <= 128.
// Block = 16 RT rows x all of H x one of gridDim.y ranges of x.  Every wave streams the weight rows of its h-tiles for
// every x of the range straight from global memory (16-byte fragment reads through a register ring), so the cost per MFMA
// of that stream falls with RT; the x split keeps the block count up when RT grows.  With two ranges the two partial sums
// are added into a zeroed `out` with atomics -- 0 + a + b is the same bits in either order.
template <bool F32IN, int KCH, int RT>
__global__ __launch_bounds__(64 * kTriWaves) void tri_kernel(const typename MfmaCfg<F32IN>::T* __restrict__ c,
                                                             const typename MfmaCfg<F32IN>::T* __restrict__ w,
                                                             const typename MfmaCfg<F32IN>::T* __restrict__ p, int M, int X,
                                                             int H, int xs, float* __restrict__ out) {
    using C = MfmaCfg<F32IN>;
    using Frag = typename C::Frag;
    constexpr int Y = KCH * C::KW, FPK = C::KW / C::EPL;   // fragment stride between K chunks, in Frag units
    constexpr int ROWS = 16 * RT;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* cT = reinterpret_cast<float*>(smem_raw);   // [x range][ROWS]: c of this block's rows, transposed, fp32
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    // 1-D grid; blocks whose ids agree modulo 8 share an XCD (and its 4 MB L2).  With two x ranges, ids {0..3} + 8k take the
    // first range and {4..7} + 8k the second: an XCD streams only ITS half of the weights (2 MB bf16 at 128^3), which then
    // stays L2-resident next to the blocks' own rows instead of thrashing a cache of exactly the weights' size.
    const int n_rb = (M + ROWS - 1) / ROWS;
    int rb, yb;
    if (xs == 2) {
        const int grp = blockIdx.x >> 3, in8 = blockIdx.x & 7;
        yb = in8 >> 2;
        rb = grp * 4 + (in8 & 3);
        if (rb >= n_rb) return;   // the grid is padded to a multiple of 8 (whole block exits: no barrier is skipped by part of it)
    } else {
        rb = blockIdx.x % n_rb;
        yb = blockIdx.x / n_rb;
    }
    const int m0 = rb * ROWS;
    const int xper = (X + xs - 1) / xs, xb = yb * xper, xe = min(X, xb + xper), nx = xe - xb;
    for (int i = threadIdx.x; i < ROWS * nx; i += 64 * kTriWaves) {
        const int row = i / nx, x = i - row * nx;   // coalesced read, transposed write
        cT[x * ROWS + row] = m0 + row < M ? tri_ld<F32IN>(c, (size_t)(m0 + row) * X + xb + x) : 0.f;
    }
    // A operand: this block's p rows, resident for the whole x loop
    Frag pf[RT][KCH];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const Frag* rowp = reinterpret_cast<const Frag*>(p + (size_t)min(m0 + 16 * rt + r, M - 1) * Y + C::EPL * g);
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) pf[rt][kc] = rowp[kc * FPK];
    }
    const int n_ht = H >> 4;
    // B operand of (x, j): rows h = 16 (wave + 4 j) + r of w[x], clamped (tiles past H are never stored)
    auto load_w = [&](int x, Frag (&f)[kTriHPW][KCH]) {
#pragma unroll
        for (int j = 0; j < kTriHPW; ++j) {
            const int ht = min(wave + kTriWaves * j, n_ht - 1);
            const Frag* rowp = reinterpret_cast<const Frag*>(w + ((size_t)x * H + 16 * ht + r) * Y + C::EPL * g);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) f[j][kc] = rowp[kc * FPK];
        }
    };
    f32x4 acc[RT][kTriHPW];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < kTriHPW; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // weight rows run through a register ring PF x-planes deep: one plane's MFMAs do not cover an L2 round trip
    constexpr int PF = (F32IN || RT > 4) ? 2 : 4;
    Frag wring[PF][kTriHPW][KCH];
#pragma unroll
    for (int u = 0; u < PF; ++u) load_w(min(xb + u, xe - 1), wring[u]);
    __syncthreads();
    for (int x0 = xb; x0 < xe; x0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int x = min(x0 + u, xe - 1);
            const float live = x0 + u < xe ? 1.f : 0.f;   // the tail of a partial ring pass contributes nothing
            float4 cv[RT];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) cv[rt] = *reinterpret_cast<const float4*>(cT + (x - xb) * ROWS + 16 * rt + 4 * g);
#pragma unroll
            for (int j = 0; j < kTriHPW; ++j)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KCH; ++kc) d = mma_chunk<F32IN>(pf[rt][kc], wring[u][j][kc], d);
                    acc[rt][j][0] = fmaf(cv[rt].x * live, d[0], acc[rt][j][0]);   // rows 4g + n: c[m, x] scales row m
                    acc[rt][j][1] = fmaf(cv[rt].y * live, d[1], acc[rt][j][1]);
                    acc[rt][j][2] = fmaf(cv[rt].z * live, d[2], acc[rt][j][2]);
                    acc[rt][j][3] = fmaf(cv[rt].w * live, d[3], acc[rt][j][3]);
                }
            __builtin_amdgcn_sched_barrier(0);
            load_w(min(x0 + u + PF, xe - 1), wring[u]);   // refill this slot: PF - 1 planes of MFMAs until it is needed
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const bool split = xs > 1;
#pragma unroll
    for (int j = 0; j < kTriHPW; ++j) {
        const int ht = wave + kTriWaves * j;
        if (ht < n_ht)
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const int m = m0 + 16 * rt + 4 * g + n;
                    if (m < M) {
                        float* o = out + (size_t)m * H + 16 * ht + r;
                        if (split) atomicAdd(o, acc[rt][j][n]);
                        else *o = acc[rt][j][n];
                    }
                }
    }
}

// ---- round 3: the same contraction with the weight planes staged through LDS (bf16, H = Y = 128) -----------------------------
// tri_kernel keeps 96 rows of p fragments and a 4-plane weight ring in 384 registers: one wave per SIMD, every weight-fragment
// read's L2 latency exposed (78 us at M = 10 496, 22 % of the bf16 matrix-core peak), and every 96-row block streams its half
// of the 4 MB weight tensor from L2: 440 MB per launch at the ~5.5 TB/s the L2 -> CU fabric delivered in every variant measured
// (a 128-row LDS-staged version moved 344 MB in 72 us whatever its pipelining or XCD mapping: the traffic, not the schedule, set
// the time).  Here a workgroup owns 256 rows x all 128 h for a range of x -- 41 row blocks x 4 MB = 164 MB per launch -- each
// weight plane w[x] (128 x 128 bf16 = 32 KB, contiguous) is loaded ONCE per workgroup, coalesced, into a double-buffered
// XOR-swizzled LDS image (16-byte segment s of row r at slot 16 r + (s ^ (r & 15)): the ds_read_b128 fragment reads are
// conflict-free without padding -- align_max_kernel's scheme) while the previous plane's MFMAs run; 8 waves, each 64 rows x 64 h
// (p fragments resident: 64 VGPRs), two per SIMD; one barrier per plane.  c[m, x] scales the plane's result tile row-wise as
// before.  The x range is split over `xs` workgroups (6 at M = 10 496: 246 workgroups); their partial sums go to separate slabs that
// a second launch adds in a fixed order (bit-reproducible; two-addend atomics when the caller has no workspace).
constexpr int kTri2Threads = 512, kTri2Rows = 256;

__global__ __launch_bounds__(kTri2Threads) void tri2_kernel(const uint16_t* __restrict__ c, const uint16_t* __restrict__ w,
                                                            const uint16_t* __restrict__ p, int M, int X, int xs,
                                                            float* __restrict__ out, size_t part_stride, int use_atomic) {
    constexpr int H = 128, Y = 128, KCH = 4;
    extern __shared__ __attribute__((aligned(16))) char tri2_smem[];
    uint4* wbuf = reinterpret_cast<uint4*>(tri2_smem);                       // [2][128 rows][16 segments]
    float* cT = reinterpret_cast<float*>(tri2_smem + 2 * H * Y * 2);        // [x range][256 rows] fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
    const int r = lane & 15, g = lane >> 4;
    const int n_rb = (M + kTri2Rows - 1) / kTri2Rows;
    const int rb = blockIdx.x % n_rb, yb = blockIdx.x / n_rb;
    const int m0 = rb * kTri2Rows;
    const int xper = (X + xs - 1) / xs, xb = yb * xper, xe = min(X, xb + xper), nx = max(xe - xb, 0);
    for (int i = tid; i < kTri2Rows * nx; i += kTri2Threads) {   // c of this block's rows, transposed, fp32
        const int row = i / nx, x = i - row * nx;
        cT[x * kTri2Rows + row] = m0 + row < M ? __uint_as_float((uint32_t)c[(size_t)(m0 + row) * X + xb + x] << 16) : 0.f;
    }
    // A operand: this wave's 64 rows of p, resident for the whole x loop
    bf16x8 pf[4][KCH];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const bf16x8* rowp = reinterpret_cast<const bf16x8*>(p + (size_t)min(m0 + wm * 64 + 16 * rt + r, M - 1) * Y + 8 * g);
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) pf[rt][kc] = rowp[kc * 4];
    }
    // staging of one weight plane: 2048 segments of 16 bytes, 4 per thread (slot: row sl >> 4, segment swizzled by the row)
    uint4 rw0, rw1, rw2, rw3;
    const int so0 = ((tid + 0 * kTri2Threads) >> 4) * 16 + (((tid + 0 * kTri2Threads) & 15) ^ (((tid + 0 * kTri2Threads) >> 4) & 15));
    const int so1 = ((tid + 1 * kTri2Threads) >> 4) * 16 + (((tid + 1 * kTri2Threads) & 15) ^ (((tid + 1 * kTri2Threads) >> 4) & 15));
    const int so2 = ((tid + 2 * kTri2Threads) >> 4) * 16 + (((tid + 2 * kTri2Threads) & 15) ^ (((tid + 2 * kTri2Threads) >> 4) & 15));
    const int so3 = ((tid + 3 * kTri2Threads) >> 4) * 16 + (((tid + 3 * kTri2Threads) & 15) ^ (((tid + 3 * kTri2Threads) >> 4) & 15));
#define VLG_TRI2_FETCH(x)                                                                     \
    do {                                                                                      \
        const uint4* src_ = reinterpret_cast<const uint4*>(w + (size_t)(x) * H * Y) + tid;    \
        rw0 = src_[0]; rw1 = src_[kTri2Threads]; rw2 = src_[2 * kTri2Threads]; rw3 = src_[3 * kTri2Threads]; \
    } while (0)
#define VLG_TRI2_STASH(buf_)                                                                  \
    do {                                                                                      \
        uint4* dst_ = wbuf + (buf_) * (H * 16);                                               \
        dst_[so0] = rw0; dst_[so1] = rw1; dst_[so2] = rw2; dst_[so3] = rw3;                   \
    } while (0)
    f32x4 acc[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nx > 0) {
        VLG_TRI2_FETCH(xb);
        VLG_TRI2_STASH(0);
        if (nx > 1) VLG_TRI2_FETCH(xb + 1);
    }
    __syncthreads();
    for (int xi = 0, buf = 0; xi < nx; ++xi, buf ^= 1) {
        const uint4* wb = wbuf + buf * (H * 16);
        float4 cv[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) cv[rt] = *reinterpret_cast<const float4*>(cT + xi * kTri2Rows + wm * 64 + 16 * rt + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int hrow = wn * 64 + j * 16 + r;   // B operand: row h of the plane, segments kc * 4 + g
            bf16x8 bfr[KCH];
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) bfr[kc] = __builtin_bit_cast(bf16x8, wb[hrow * 16 + ((kc * 4 + g) ^ (hrow & 15))]);
            f32x4 d[4];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) d[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
                for (int rt = 0; rt < 4; ++rt) d[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[rt][kc], bfr[kc], d[rt], 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                acc[rt][j][0] = fmaf(cv[rt].x, d[rt][0], acc[rt][j][0]);
                acc[rt][j][1] = fmaf(cv[rt].y, d[rt][1], acc[rt][j][1]);
                acc[rt][j][2] = fmaf(cv[rt].z, d[rt][2], acc[rt][j][2]);
                acc[rt][j][3] = fmaf(cv[rt].w, d[rt][3], acc[rt][j][3]);
            }
        }
        if (xi + 1 < nx) {
            VLG_TRI2_STASH(buf ^ 1);
            if (xi + 2 < nx) VLG_TRI2_FETCH(xb + xi + 2);
        }
        __syncthreads();
    }
#undef VLG_TRI2_FETCH
#undef VLG_TRI2_STASH
    float* dstb = out + (use_atomic ? 0 : (size_t)yb * part_stride);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int m = m0 + wm * 64 + 16 * rt + 4 * g + n;
                if (m < M) {
                    float* o = dstb + (size_t)m * H + wn * 64 + j * 16 + r;
                    if (use_atomic) atomicAdd(o, acc[rt][j][n]);   // two ranges: 0 + a + b is the same bits in either order
                    else *o = acc[rt][j][n];
                }
            }
}

// out[i] = sum_s part[s][i], s ascending (n % 4 == 0)
__global__ __launch_bounds__(256) void tri2_reduce_kernel(const float* __restrict__ part, int S, size_t n, float* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 t = *reinterpret_cast<const float4*>(part + i);
    for (int s = 1; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * n + i);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(out + i) = t;
}

static bool tri2_applies(int M, int X, int H, int Y, bool f32in) { return !f32in && H == 128 && Y == 128 && M >= 1024 && X >= 2 && X <= 256; }
static int tri2_splits(int M, int X) {   // ~one workgroup per CU
    const int n_rb = (M + kTri2Rows - 1) / kTri2Rows;
    int xs = (256 + n_rb / 2) / n_rb;
    xs = xs < 1 ? 1 : (xs > 8 ? 8 : xs);
    while (xs > 1 && X / xs < 8) --xs;
    return xs;
}
static size_t tri2_part_bytes(int M, int X) { const int xs = tri2_splits(M, X); return xs > 1 ? sizeof(float) * (size_t)xs * M * 128 : 0; }

// part: xs slabs of [M][128] floats (fixed-order sum into out), or null: two ranges met by atomicAdd in a zeroed out
static int launch_tri2(const void* c, const void* w, const void* p, int M, int X, float* out, float* part, hipStream_t s) {
    const int xs = part ? tri2_splits(M, X) : 2;
    const int n_rb = (M + kTri2Rows - 1) / kTri2Rows;
    const size_t lds = 2 * (size_t)128 * 128 * 2 + sizeof(float) * kTri2Rows * ((X + xs - 1) / xs);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    const bool atomic = part == nullptr || xs == 1;
    if (atomic && xs > 1) {
        e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)M * 128, s);
        if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(tri2_kernel, dim3(n_rb * xs), dim3(kTri2Threads), lds, s, (const uint16_t*)c, (const uint16_t*)w, (const uint16_t*)p,
                       M, X, xs, atomic ? out : part, (size_t)M * 128, (atomic && xs > 1) ? 1 : 0);
    if (!atomic) {
        const size_t n = (size_t)M * 128;
        hipLaunchKernelGGL(tri2_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, part, xs, n, out);
    }
    return 0;
}

// ---- round 5: float32 operands on the fp16 matrix cores (H = Y = 128) -- the reference's `precision: 32`, config/trainer/train.yaml:20 ----
// tri_kernel<fp32> runs on v_mfma_f32_16x16x4_f32 (1/16 of the 16-bit rate): 3 x 0.43 ms + 0.37 ms per training step.  Here every float32
// row r of an MFMA operand is written as two fp16 rows, v s_r = hi + lo + e with hi = fp16(v s_r), lo = fp16(v s_r - hi), s_r the power of two
// that puts the row's largest magnitude in [2^14, 2^15): |e| <= max(2^-23 |v s_r|, 2^-25), i.e. 22 bits of every element within 2^-17 of its
// row's maximum and an ABSOLUTE floor of 2^-39 of that maximum below.  A product is then hi hi + hi lo + lo hi (three v_mfma_f32_16x16x32_f16
// into one fp32 accumulator), the dropped lo lo term is <= 2^-22 of it -- float32's own rounding level, where two bf16 parts (8 + 8 bits)
// stop at 2^-16 (round 4 measured that form: 2e-5 against the 6e-6 the float32 tests hold).  Powers of two scale exactly; 1 / s_r of the p
// rows goes into the per-row factor c[m,x] (float32, never split), 1 / s of the weight rows (x, h) scales the plane's result column-wise.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ void split_f16_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = v - __builtin_convertvector(h, f32x2);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, f16x2));
}
// the power of two s with m s in [2^14, 2^15) (m > 0, normal), as (s, 1 / s); 1 for m = 0
__device__ __forceinline__ void pow2_scale(float m, float& sc, float& inv) {
    int k = m > 0.f ? 14 - ((int)((__float_as_uint(m) >> 23) & 0xffu) - 127) : 0;
    k = k < -110 ? -110 : (k > 110 ? 110 : k);
    sc = __uint_as_float((uint32_t)(127 + k) << 23);
    inv = __uint_as_float((uint32_t)(127 - k) << 23);
}

// rows of 128 float32 -> hi / lo fp16 planes of the scaled row and inv[row] = 1 / s_row.  16 lanes per row, 16 rows per workgroup.
__global__ __launch_bounds__(256) void tri_split_rows_kernel(const float* __restrict__ src, size_t rows, uint16_t* __restrict__ hi,
                                                             uint16_t* __restrict__ lo, float* __restrict__ inv) {
    const size_t row = (size_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15;
    const bool ok = row < rows;
    const size_t rr = ok ? row : rows - 1;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + rr * 128 + sub * 8), b = *reinterpret_cast<const f32x4*>(src + rr * 128 + sub * 8 + 4);
    float m = fmaxf(fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3]))), fmaxf(fmaxf(fabsf(b[0]), fabsf(b[1])), fmaxf(fabsf(b[2]), fabsf(b[3]))));
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 16));
    float sc, iv;
    pow2_scale(m, sc, iv);
    uint4 h, l;
    split_f16_pair(a[0] * sc, a[1] * sc, h.x, l.x);
    split_f16_pair(a[2] * sc, a[3] * sc, h.y, l.y);
    split_f16_pair(b[0] * sc, b[1] * sc, h.z, l.z);
    split_f16_pair(b[2] * sc, b[3] * sc, h.w, l.w);
    if (ok) {
        *reinterpret_cast<uint4*>(hi + row * 128 + sub * 8) = h;
        *reinterpret_cast<uint4*>(lo + row * 128 + sub * 8) = l;
        if (sub == 0) inv[row] = iv;
    }
}

// tri2_kernel's scheme on two fp16 parts per operand: a workgroup owns 256 rows x all 128 h for a range of x; wave w holds the hi / lo
// fragments of ITS 32 rows of p for the whole x loop (64 VGPRs) and all eight h tiles (64 accumulator registers).  The weights are staged
// HALF a plane at a time (64 h rows, hi | lo: 32 KB, double-buffered; their 64 inverse row scales behind them): 16 staging registers per
// lane instead of 32 -- with whole planes the kernel needed 270 registers at the 256 that two wavefronts per SIMD leave.
constexpr int kTri3Threads = 512, kTri3Rows = 256;

__global__ __launch_bounds__(kTri3Threads) void tri3_kernel(const float* __restrict__ c, const uint16_t* __restrict__ wh,
                                                            const uint16_t* __restrict__ wl, const float* __restrict__ winv,
                                                            const uint16_t* __restrict__ ph, const uint16_t* __restrict__ pl,
                                                            const float* __restrict__ pinv, int M, int X, int xs, float* __restrict__ out,
                                                            size_t part_stride, int use_atomic) {
    constexpr int H = 128, Y = 128, KCH = 4, HALF = 64 * 16;   // half a plane part: 64 rows x 16 segments of 16 bytes
    extern __shared__ __attribute__((aligned(16))) char tri3_smem[];
    uint4* wbuf = reinterpret_cast<uint4*>(tri3_smem);                                   // [2 buffers][hi | lo][64 rows][16 segments]
    float* wis = reinterpret_cast<float*>(tri3_smem + 4 * (size_t)HALF * 16);             // [2 buffers][64] inverse row scales
    float* cT = wis + 2 * 64;                                                             // [x range][256 rows]: c[m,x] / s_p[m]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n_rb = (M + kTri3Rows - 1) / kTri3Rows;
    const int rb = blockIdx.x % n_rb, yb = blockIdx.x / n_rb;
    const int m0 = rb * kTri3Rows;
    const int xper = (X + xs - 1) / xs, xb = yb * xper, xe = min(X, xb + xper), nx = max(xe - xb, 0);
    for (int i = tid; i < kTri3Rows * nx; i += kTri3Threads) {
        const int row = i / nx, x = i - row * nx;
        cT[x * kTri3Rows + row] = m0 + row < M ? c[(size_t)(m0 + row) * X + xb + x] * pinv[m0 + row] : 0.f;
    }
    f16x8 pfh[2][KCH], pfl[2][KCH];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const size_t off = (size_t)min(m0 + wave * 32 + 16 * rt + r, M - 1) * Y + 8 * g;
        const f16x8 *rh = reinterpret_cast<const f16x8*>(ph + off), *rl = reinterpret_cast<const f16x8*>(pl + off);
#pragma unroll
        for (int kc = 0; kc < KCH; ++kc) {
            pfh[rt][kc] = rh[kc * 4];
            pfl[rt][kc] = rl[kc * 4];
        }
    }
    uint4 rwh0, rwh1, rwl0, rwl1;   // (scalars: as arrays captured by the lambdas they stayed in scratch memory)
    float rwi = 0.f;
    const int so0 = (tid >> 4) * 16 + ((tid & 15) ^ ((tid >> 4) & 15)), so1 = so0 + 32 * 16;   // (row + 32: the same row & 15)
    auto fetch = [&](int t) {   // stage t = (x, half)
        const size_t seg = (size_t)(xb + (t >> 1)) * (H * 16) + (size_t)(t & 1) * HALF + tid;
        const uint4 *sh = reinterpret_cast<const uint4*>(wh) + seg, *sl = reinterpret_cast<const uint4*>(wl) + seg;
        rwh0 = sh[0]; rwh1 = sh[kTri3Threads];
        rwl0 = sl[0]; rwl1 = sl[kTri3Threads];
        if (tid < 64) rwi = winv[(size_t)(xb + (t >> 1)) * H + (t & 1) * 64 + tid];
    };
    auto stash = [&](int buf) {
        uint4* dh = wbuf + buf * (2 * HALF);
        dh[so0] = rwh0; dh[so1] = rwh1;
        dh[HALF + so0] = rwl0; dh[HALF + so1] = rwl1;
        if (tid < 64) wis[buf * 64 + tid] = rwi;
    };
    f32x4 acc[2][8];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[rt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nt = 2 * nx;
    if (nt > 0) {
        fetch(0);
        stash(0);
        fetch(1);
    }
    __syncthreads();
    int foff[KCH];
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) foff[kc] = r * 16 + ((kc * 4 + g) ^ r);
    for (int xi = 0; xi < nx; ++xi) {
        f32x4 cv[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) cv[rt] = *reinterpret_cast<const f32x4*>(cT + xi * kTri3Rows + wave * 32 + 16 * rt + 4 * g);
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // stage 2 xi + half sits in buffer `half`
            const uint4* wb = wbuf + half * (2 * HALF);
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4) {
                const float wi = wis[half * 64 + j4 * 16 + r];
                f32x4 d[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    const f16x8 bh = __builtin_bit_cast(f16x8, wb[j4 * 256 + foff[kc]]), bl = __builtin_bit_cast(f16x8, wb[HALF + j4 * 256 + foff[kc]]);
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {   // the two small terms first
                        d[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfl[rt][kc], bh, d[rt], 0, 0, 0);
                        d[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfh[rt][kc], bl, d[rt], 0, 0, 0);
                        d[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pfh[rt][kc], bh, d[rt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[rt][half * 4 + j4][n] = fmaf(cv[rt][n] * wi, d[rt][n], acc[rt][half * 4 + j4][n]);
                __builtin_amdgcn_sched_barrier(0);   // (left alone, hipcc hoists the 32 fragment reads of a stage to its top: 128 more registers, scratch)
            }
            const int t = 2 * xi + half;
            if (t + 1 < nt) {
                stash(half ^ 1);
                if (t + 2 < nt) fetch(t + 2);
            }
            __syncthreads();
        }
    }
    float* dstb = out + (use_atomic ? 0 : (size_t)yb * part_stride);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int m = m0 + wave * 32 + 16 * rt + 4 * g + n;
                if (m < M) {
                    float* o = dstb + (size_t)m * H + j * 16 + r;
                    if (use_atomic) atomicAdd(o, acc[rt][j][n]);
                    else *o = acc[rt][j][n];
                }
            }
}

static bool tri3_applies(int M, int X, int H, int Y) { return H == 128 && Y == 128 && M >= 1024 && X >= 2 && X <= 256; }
static int tri3_splits(int M, int X) {   // tri2's; the x range's [range][256] float32 factors sit behind the 64 KB of weight buffers
    int xs = tri2_splits(M, X);
    while ((X + xs - 1) / xs > 80) ++xs;
    return xs;
}
static size_t tri3_part_bytes(int M, int X) { const int xs = tri3_splits(M, X); return xs > 1 ? sizeof(float) * (size_t)xs * M * 128 : 0; }

struct Tri3Ops {   // one operand in two fp16 parts + the inverse row scales
    uint16_t *hi, *lo;
    float* inv;
};
// carving of a [rows][128] operand's split form out of a scratch region; returns the bytes used (256-byte granules)
static size_t tri3_carve(char* base, size_t rows, Tri3Ops* o) {
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t plane = up(rows * 128 * 2), iv = up(rows * 4);
    if (o) {
        o->hi = reinterpret_cast<uint16_t*>(base);
        o->lo = reinterpret_cast<uint16_t*>(base + plane);
        o->inv = reinterpret_cast<float*>(base + 2 * plane);
    }
    return 2 * plane + iv;
}
static void launch_split_rows(const float* src, size_t rows, const Tri3Ops& o, hipStream_t s) {
    hipLaunchKernelGGL(tri_split_rows_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, src, rows, o.hi, o.lo, o.inv);
}

// out[m,h] = sum_x c[m,x] sum_y w[x,h,y] p[m,y] from the split forms of w ([X 128 rows]) and p ([M rows]); part: tri3_part_bytes(M, X)
static int launch_tri3(const float* c, const Tri3Ops& w, const Tri3Ops& p, int M, int X, float* out, float* part, hipStream_t s) {
    const int xs = tri3_splits(M, X);
    const int n_rb = (M + kTri3Rows - 1) / kTri3Rows;
    const size_t lds = 4 * (size_t)64 * 128 * 2 + sizeof(float) * (2 * 64 + (size_t)kTri3Rows * ((X + xs - 1) / xs));
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(tri3_kernel, dim3(n_rb * xs), dim3(kTri3Threads), lds, s, c, w.hi, w.lo, w.inv, p.hi, p.lo, p.inv, M, X, xs, xs > 1 ? part : out,
                       (size_t)M * 128, 0);
    if (xs > 1) {
        const size_t n = (size_t)M * 128;
        hipLaunchKernelGGL(tri2_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, part, xs, n, out);
    }
    return 0;
}

// workspace of the float32 forward: split w | split p | partial slabs
static size_t tri3_fwd_bytes(int M, int X) {
    return tri3_carve(nullptr, (size_t)X * 128, nullptr) + tri3_carve(nullptr, (size_t)M, nullptr) + ((tri3_part_bytes(M, X) + 255) & ~(size_t)255);
}
static int launch_tri3_forward(const void* c, const void* w, const void* p, int M, int X, float* out, char* ws, hipStream_t s) {
    Tri3Ops wo, po;
    size_t off = tri3_carve(ws, (size_t)X * 128, &wo);
    off += tri3_carve(ws + off, (size_t)M, &po);
    launch_split_rows((const float*)w, (size_t)X * 128, wo, s);
    launch_split_rows((const float*)p, (size_t)M, po, s);
    return launch_tri3((const float*)c, wo, po, M, X, out, reinterpret_cast<float*>(ws + off), s);
}

template <bool F32IN, int KCH, int RT>
static int launch_tri_rt(const void* c, const void* w, const void* p, int M, int X, int H, int xs, float* out, hipStream_t s) {
    using T = typename MfmaCfg<F32IN>::T;
    const size_t lds = sizeof(float) * 16 * RT * ((X + xs - 1) / xs);
    auto k = tri_kernel<F32IN, KCH, RT>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    }
    if (xs > 1) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)M * H, s);
        if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    const int n_rb = (M + 16 * RT - 1) / (16 * RT);
    const int blocks = xs == 2 ? ((n_rb + 3) / 4) * 8 : n_rb * xs;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * kTriWaves), lds, s, (const T*)c, (const T*)w, (const T*)p, M, X, H, xs, out);
    return 0;
}

static int device_cus() {
    static int n = 0;   // one device per process (the launch contract); 256 on MI355X
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

// Rows per block.  The timings below behave as if a CU ran one four-wave block at a time: the launch takes
// ceil(blocks / CUs) rounds of (weight stream of the x range, constant) + (MFMA work, ~ rows).  Fitted on M = 10 496:
// bf16 is stream-heavy (76 us + 2.2 us per 16 rows: 96 rows x 2 ranges = 220 blocks in one round, 90 us, against
// 94 us for 128 rows and 156 us for 80 rows = 264 blocks = two rounds); fp32 is MFMA-heavy (75 + 51 us per 16 rows:
// 48 rows x 2 = 438 blocks, 422 us, against 164 blocks of 64 rows on 256 CUs, 512 us).
template <bool F32IN, int KCH>
static int launch_tri(const void* c, const void* w, const void* p, int M, int X, int H, float* out, hipStream_t s) {
    const int xs = X >= 32 ? 2 : 1;
    const int cus = device_cus();
    const float stream = F32IN ? 1.5f : 34.f;   // the constant term in units of one 16-row MFMA pass
    const int cand[3] = {F32IN ? 3 : 4, F32IN ? 4 : 6, F32IN ? 4 : 6};   // (bf16 with 8 row tiles needs 512 registers + scratch: dropped)
    int best = cand[2];
    float best_cost = 0.f;
    for (int i = 0; i < 3; ++i) {
        const long blocks = (long)((M + 16 * cand[i] - 1) / (16 * cand[i])) * xs;
        const float cost = (float)((blocks + cus - 1) / cus) * ((float)cand[i] + stream);
        if (i == 0 || cost < best_cost) { best = cand[i]; best_cost = cost; }
    }
    if constexpr (F32IN) {
        if (best == 3) return launch_tri_rt<true, KCH, 3>(c, w, p, M, X, H, xs, out, s);
        return launch_tri_rt<true, KCH, 4>(c, w, p, M, X, H, xs, out, s);
    } else {
        if (best == 4) return launch_tri_rt<false, KCH, 4>(c, w, p, M, X, H, xs, out, s);
        return launch_tri_rt<false, KCH, 6>(c, w, p, M, X, H, xs, out, s);
    }
}

// Y (the contracted, memory-contiguous dimension) decides the instantiation.
static int dispatch_tri(const void* c, const void* w, const void* p, int M, int X, int H, int Y, bool f32in, float* out,
                        hipStream_t s, float* part = nullptr) {
    if (f32in) {
        int rc;
        if (part && tri3_applies(M, X, H, Y) && !VLG_ENV("VLG_TRI_F32_EXACT")) rc = launch_tri3_forward(c, w, p, M, X, out, reinterpret_cast<char*>(part), s);   // (part: tri3_fwd_bytes)
        else if (Y == 128) rc = launch_tri<true, 8>(c, w, p, M, X, H, out, s);
        else if (Y == 64) rc = launch_tri<true, 4>(c, w, p, M, X, H, out, s);
        else if (Y == 32) rc = launch_tri<true, 2>(c, w, p, M, X, H, out, s);
        else return set_error(VLG_ERR_SHAPE, "trilinear: contracted dimension %d (supported: 32, 64, 128)", Y);
        if (rc) return rc;
    } else {
        int rc;
        if (tri2_applies(M, X, H, Y, false)) rc = launch_tri2(c, w, p, M, X, out, part, s);
        else if (Y == 128) rc = launch_tri<false, 4>(c, w, p, M, X, H, out, s);
        else if (Y == 64) rc = launch_tri<false, 2>(c, w, p, M, X, H, out, s);
        else if (Y == 32) rc = launch_tri<false, 1>(c, w, p, M, X, H, out, s);
        else return set_error(VLG_ERR_SHAPE, "trilinear: contracted dimension %d (supported: 32, 64, 128)", Y);
        if (rc) return rc;
    }
    return check_launch("tri_kernel");
}

// ---- helpers for the adjoint ----------------------------------------------------------------------------------------
// dst[a][b][c] (in T) = src[x][h][y] with (a, b, c) a permutation of (x, h, y):  mode 0: [y][x][h],  mode 1: [x][y][h].
template <typename T>
__global__ __launch_bounds__(256) void tri_permute_kernel(const T* __restrict__ src, T* __restrict__ dst, int X, int H, int Y,
                                                          int mode) {
    const size_t n = (size_t)X * H * Y;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int h = (int)(i % H);   // dst index = ((a * B + b) * H + h): h is the fastest dimension in both modes
        const size_t ab = i / H;
        int x, y;
        if (mode == 0) { x = (int)(ab % X); y = (int)(ab / X); }
        else { y = (int)(ab % Y); x = (int)(ab / Y); }
        dst[i] = src[((size_t)x * H + h) * Y + y];
    }
}

// both permutations from one read of the weights: dA[y][x][h] = dB[x][y][h] = src[x][h][y].  A block = one x and a 32-wide strip of y,
// transposed through LDS so that the reads (y fastest) and both writes (h fastest) are coalesced.
template <typename T>
__global__ __launch_bounds__(256) void tri_permute2_kernel(const T* __restrict__ src, T* __restrict__ dA, T* __restrict__ dB, int X, int H,
                                                           int Y) {
    __shared__ T tile[32][130];   // [y in strip][h], H <= 128
    const int x = blockIdx.x, y0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < H * 32; i += 256) {
        const int h = i >> 5, yy = i & 31;
        if (y0 + yy < Y) tile[yy][h] = src[((size_t)x * H + h) * Y + y0 + yy];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * H; i += 256) {
        const int yy = i / H, h = i - yy * H;
        if (y0 + yy < Y) {
            const T v = tile[yy][h];
            dA[((size_t)(y0 + yy) * X + x) * H + h] = v;
            dB[((size_t)x * Y + y0 + yy) * H + h] = v;
        }
    }
}

// dst[d][m] (in T, row pitch Mp, zero beyond M) = src[m][d];  CAST: src is fp32 and is rounded to T, else src is T.
template <typename T, bool CAST>
__global__ __launch_bounds__(256) void tri_transpose_kernel(const void* __restrict__ src_, T* __restrict__ dst, int M, int D,
                                                            int Mp) {
    __shared__ float tile[32][33];
    const int m0 = blockIdx.x * 32, d0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        const int m = m0 + k, d = d0 + tx;
        float v = 0.f;
        if (m < M && d < D) {
            if constexpr (CAST) v = reinterpret_cast<const float*>(src_)[(size_t)m * D + d];
            else if constexpr (sizeof(T) == 4) v = reinterpret_cast<const float*>(src_)[(size_t)m * D + d];
            else v = __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(src_)[(size_t)m * D + d] << 16);
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int d = d0 + k, m = m0 + tx;
        if (d < D && m < Mp) {
            const float v = tile[tx][k];
            if constexpr (sizeof(T) == 4) dst[(size_t)d * Mp + m] = v;
            else {   // round to nearest even bf16
                const uint32_t u = __float_as_uint(v);
                dst[(size_t)d * Mp + m] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
        }
    }
}

// row-major copy of the fp32 cotangent in the operand type
__global__ __launch_bounds__(256) void tri_cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t u = __float_as_uint(src[i]);
        dst[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
}

// d_w[x,h,y] = sum_m c[m,x] g[m,h] p[m,y] from the transposed copies cT [X][Mp], gT [H][Mp], pT [Y][Mp] (m contiguous, zero
// padded to Mp, a multiple of the K chunk).  Wave = (x, a group of kDwHT h-tiles) x all y-tiles; the A operand of chunk
// k is gT[h][k..] * cT[x][k..] formed in registers (for bf16: product in fp32, rounded back to bf16).
constexpr int kDwHT = 4;   // h-tiles per wave: their A fragments reuse every p fragment
constexpr int kDwYT = 8;   // y-tiles: Y <= 128

template <bool F32IN>
__device__ __forceinline__ typename MfmaCfg<F32IN>::Frag frag_mul(const typename MfmaCfg<F32IN>::Frag& a,
                                                                const typename MfmaCfg<F32IN>::Frag& b) {
    if constexpr (F32IN) return a * b;
    else {
        typename MfmaCfg<false>::Frag o;
        const uint32_t* ua = reinterpret_cast<const uint32_t*>(&a);
        const uint32_t* ub = reinterpret_cast<const uint32_t*>(&b);
        uint32_t* uo = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // two bf16 per dword
            const float lo = __uint_as_float(ua[i] << 16) * __uint_as_float(ub[i] << 16);
            const float hi = __uint_as_float(ua[i] & 0xffff0000u) * __uint_as_float(ub[i] & 0xffff0000u);
            const uint32_t ul = __float_as_uint(lo), uh = __float_as_uint(hi);
            uo[i] = ((ul + 0x7fffu + ((ul >> 16) & 1u)) >> 16) | ((uh + 0x7fffu + ((uh >> 16) & 1u)) & 0xffff0000u);
        }
        return o;
    }
}

constexpr int kDwSplit = 4;   // waves per block: wave s takes K chunks s, s + 4, ...; partial tiles meet in LDS, fixed order

template <bool F32IN>
__global__ __launch_bounds__(64 * kDwSplit) void tri_dw_kernel(const typename MfmaCfg<F32IN>::T* __restrict__ cT,
                                                    const typename MfmaCfg<F32IN>::T* __restrict__ gT,
                                                    const typename MfmaCfg<F32IN>::T* __restrict__ pT, int Mp, int X, int H,
                                                    int Y, float* __restrict__ d_w) {
    using C = MfmaCfg<F32IN>;
    using Frag = typename C::Frag;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
    const int x = blockIdx.x, hg = blockIdx.y;   // h-tiles hg * kDwHT ..
    const int n_ht = H >> 4, n_yt = Y >> 4, n_chunk = Mp / C::KW;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f32x4* part = reinterpret_cast<f32x4*>(smem_raw);   // [kDwSplit - 1][kDwHT * kDwYT][64 lanes]
    const Frag* crow = reinterpret_cast<const Frag*>(cT + (size_t)x * Mp + C::EPL * g);   // same for every row of the A tile
    const Frag* grow[kDwHT];
    const Frag* prow[kDwYT];
#pragma unroll
    for (int j = 0; j < kDwHT; ++j)
        grow[j] = reinterpret_cast<const Frag*>(gT + (size_t)(16 * min(hg * kDwHT + j, n_ht - 1) + r) * Mp + C::EPL * g);
#pragma unroll
    for (int t = 0; t < kDwYT; ++t) prow[t] = reinterpret_cast<const Frag*>(pT + (size_t)(16 * min(t, n_yt - 1) + r) * Mp + C::EPL * g);
    constexpr int FPK = C::KW / C::EPL;
    f32x4 acc[kDwHT][kDwYT];
#pragma unroll
    for (int j = 0; j < kDwHT; ++j)
#pragma unroll
        for (int t = 0; t < kDwYT; ++t) acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // operands of the wave's chunks run through a register ring PFD chunks deep: one chunk's MFMAs (0.2 us bf16) cover a
    // fraction of an L2 round trip, and with one wave per SIMD nothing else hides it (depth 1 measured 120 us bf16)
    constexpr int PFD = F32IN ? 2 : 1;   // bf16: deeper rings spill next to the 128 accumulators and measured slower (128 vs 120 us)
    Frag cf[PFD], gf[PFD][kDwHT], pf[PFD][kDwYT];
    auto fetch = [&](int k, int slot) {   // chunk index k (clamped: the tail re-reads the last chunk, masked below)
        const int kk = min(k, n_chunk - 1) * FPK;
        cf[slot] = crow[kk];
#pragma unroll
        for (int j = 0; j < kDwHT; ++j) gf[slot][j] = grow[j][kk];
#pragma unroll
        for (int t = 0; t < kDwYT; ++t) pf[slot][t] = prow[t][kk];
    };
#pragma unroll
    for (int u = 0; u < PFD; ++u) fetch(wave + kDwSplit * u, u);
    for (int k0 = wave; k0 < n_chunk; k0 += kDwSplit * PFD) {
#pragma unroll
        for (int u = 0; u < PFD; ++u) {
            const int k = k0 + kDwSplit * u;
            if (k < n_chunk) {   // uniform
#pragma unroll
                for (int j = 0; j < kDwHT; ++j) {
                    const Frag a = frag_mul<F32IN>(gf[u][j], cf[u]);
#pragma unroll
                    for (int t = 0; t < kDwYT; ++t) acc[j][t] = mma_chunk<F32IN>(a, pf[u][t], acc[j][t]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            fetch(k + kDwSplit * PFD, u);   // refill this slot: PFD - 1 chunks of MFMAs until it is needed
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int j = 0; j < kDwHT; ++j)
#pragma unroll
            for (int t = 0; t < kDwYT; ++t) part[((wave - 1) * kDwHT * kDwYT + j * kDwYT + t) * 64 + lane] = acc[j][t];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int sp = 0; sp < kDwSplit - 1; ++sp)
#pragma unroll
        for (int j = 0; j < kDwHT; ++j)
#pragma unroll
            for (int t = 0; t < kDwYT; ++t) acc[j][t] += part[(sp * kDwHT * kDwYT + j * kDwYT + t) * 64 + lane];
#pragma unroll
    for (int j = 0; j < kDwHT; ++j) {
        const int ht = hg * kDwHT + j;
        if (ht < n_ht)
#pragma unroll
            for (int t = 0; t < kDwYT; ++t)
                if (t < n_yt)
#pragma unroll
                    for (int n = 0; n < 4; ++n)   // tile rows = h, cols = y
                        d_w[((size_t)x * H + 16 * ht + 4 * g + n) * Y + 16 * t + r] = acc[j][t][n];
    }
}

// ---- d_w on transposed LDS reads (round 3; bf16, H = Y = 128) ---------------------------------------------------------------
// d_w[x] = (g . c[:,x])^T p is a [128 x M] x [M x 128] product per x whose contraction index m is the slow dimension of all
// three operands: tri_dw_kernel above reads transposed copies (three transpose launches) and streams them from L2 per
// (x, h-group) wave at 12 % matrix-core utilisation.  Here one workgroup owns XB values of x and a chunk of the rows: each
// 64-row stage of g and p is loaded once (row-major, coalesced), the A operand g[m,:] * c[m,x] is formed ONCE per stage as
// it is written to LDS (fp32 product, rounded to bf16 like tri_dw_kernel's), and both operands reach the MFMA through
// ds_read_b64_tr_b16 (vlg_mfma.h).  Row chunks give the chip 256 workgroups; their partial tiles are added in a fixed order.
constexpr int kDw2Threads = 1024, kDw2Stage = 64, kDw2XB = 2;   // 16 waves: four per SIMD (the fragment reads' LDS latency needs them), each a 32 x 32 part of the 128 x 128 tile
constexpr int kDw2Pitch = 128 * 2 + 32;   // 288 r mod 256 = 32 r: 8 consecutive rows on 8 disjoint 8-bank groups

__device__ __forceinline__ uint32_t dw2_scale2(uint32_t v, float cf) {   // two bf16 in a dword, each times cf, rounded to nearest even
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const float lo = __uint_as_float(v << 16) * cf, hi = __uint_as_float(v & 0xffff0000u) * cf;
    const bf16x2 o = {(__bf16)lo, (__bf16)hi};   // v_cvt_pk_bf16_f32: round to nearest even, like tri_dw_kernel's integer rounding
    return __builtin_bit_cast(uint32_t, o);
}

__global__ __launch_bounds__(kDw2Threads) void tri_dw2_kernel(const uint16_t* __restrict__ c, const uint16_t* __restrict__ g,
                                                              const uint16_t* __restrict__ p, int M, int X, int KC, int S,
                                                              float* __restrict__ part) {
    constexpr int HY = 128;
    extern __shared__ __attribute__((aligned(16))) char dw2_smem[];   // two stage buffers of (XB + 1) tiles
    constexpr int kTileBytes = kDw2Stage * kDw2Pitch, kBufBytes = (kDw2XB + 1) * kTileBytes;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
    // blocks whose ids agree modulo 8 share an XCD and its L2: give them the SAME row chunk (S divides 8), so a chunk of g and
    // p is fetched from HBM / Infinity Cache once per XCD pair instead of once per workgroup
    const int s = blockIdx.x % S, x0 = (blockIdx.x / S) * kDw2XB;
    const int k_begin = s * KC, k_end = min(M, k_begin + KC), rows = k_end - k_begin;
    // staging: a tile row is 128 bf16 = 16 x 16 bytes; 1024 threads cover the 64 rows of a stage in one pass
    constexpr int NP = kDw2Stage / (kDw2Threads / 16), RPP = kDw2Threads / 16;
    const int ch = tid & 15, r0 = tid >> 4;
    uint4 rg[NP], rp[NP];
    uint32_t rc[NP];
    auto fetch = [&](int ks) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int k = k_begin + ks + r0 + RPP * q;
            const bool ok = k < k_end;
            rg[q] = ok ? *reinterpret_cast<const uint4*>(g + (size_t)k * HY + ch * 8) : make_uint4(0, 0, 0, 0);
            rp[q] = ok ? *reinterpret_cast<const uint4*>(p + (size_t)k * HY + ch * 8) : make_uint4(0, 0, 0, 0);
            rc[q] = ok ? *reinterpret_cast<const uint32_t*>(c + (size_t)k * X + x0) : 0u;   // c[k, x0], c[k, x0 + 1]
        }
    };
    auto stash = [&](int buf) {
        char* sP = dw2_smem + buf * kBufBytes + kDw2XB * kTileBytes;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int off = (r0 + RPP * q) * kDw2Pitch + ch * 16;
            *reinterpret_cast<uint4*>(sP + off) = rp[q];
#pragma unroll
            for (int xb = 0; xb < kDw2XB; ++xb) {
                const float cf = __uint_as_float(xb == 0 ? rc[q] << 16 : rc[q] & 0xffff0000u);
                uint4 o;
                o.x = dw2_scale2(rg[q].x, cf); o.y = dw2_scale2(rg[q].y, cf);
                o.z = dw2_scale2(rg[q].z, cf); o.w = dw2_scale2(rg[q].w, cf);
                *reinterpret_cast<uint4*>(dw2_smem + buf * kBufBytes + xb * kTileBytes + off) = o;
            }
        }
    };
    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int row_off = (4 * g4 + q4) * kDw2Pitch + p4 * 8;
    f32x4 acc[kDw2XB][2][2] = {};
    // one barrier per stage: while the MFMAs of stage i read buffer i & 1, stage i + 1 (in registers since the previous
    // iteration) is scaled and written to the other buffer, and stage i + 2 is requested from memory
    fetch(0);
    stash(0);
    if (kDw2Stage < rows) fetch(kDw2Stage);
    __syncthreads();
    for (int ks = 0, buf = 0; ks < rows; ks += kDw2Stage, buf ^= 1) {
        const char* base = dw2_smem + buf * kBufBytes;
#pragma unroll
        for (int kk = 0; kk < kDw2Stage / 32; ++kk) {
            bf16x8 fb[2];
            const char* pb = base + kDw2XB * kTileBytes + row_off + kk * 32 * kDw2Pitch + (wn * 32) * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j] = tr_frag(tr_read(pb + j * 32), tr_read(pb + 16 * kDw2Pitch + j * 32));
#pragma unroll
            for (int xb = 0; xb < kDw2XB; ++xb) {
                const char* pa = base + xb * kTileBytes + row_off + kk * 32 * kDw2Pitch + (wm * 32) * 2;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const bf16x8 fa = tr_frag(tr_read(pa + i * 32), tr_read(pa + 16 * kDw2Pitch + i * 32));
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[xb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb[j], acc[xb][i][j], 0, 0, 0);
                }
            }
        }
        if (ks + kDw2Stage < rows) {
            stash(buf ^ 1);
            if (ks + 2 * kDw2Stage < rows) fetch(ks + 2 * kDw2Stage);
        }
        __syncthreads();
    }
#pragma unroll
    for (int xb = 0; xb < kDw2XB; ++xb) {
        if (x0 + xb >= X) break;
        float* out = part + ((size_t)s * X + x0 + xb) * HY * HY;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(size_t)(wm * 32 + i * 16 + 4 * g4 + r) * HY + wn * 32 + j * 16 + (lane & 15)] = acc[xb][i][j][r];
    }
}

// d_w[i] = sum_s part[s][i], s ascending (n % 4 == 0)
__global__ __launch_bounds__(256) void tri_dw2_reduce_kernel(const float* __restrict__ part, int S, size_t n, float* __restrict__ d_w) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    float4 t = *reinterpret_cast<const float4*>(part + i);
    for (int s = 1; s < S; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(part + (size_t)s * n + i);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    *reinterpret_cast<float4*>(d_w + i) = t;
}

static bool dw2_applies(int M, int X, int H, int Y, bool f32in) { return !f32in && H == 128 && Y == 128 && X % kDw2XB == 0 && M >= 1024; }
static int dw2_splits(int M, int X) {   // 1, 2, 4 or 8 (a divisor of the XCD count): ~one workgroup per CU, at least four stages each
    int S = 8;
    while (S > 1 && ((X / kDw2XB) * (S / 2) >= 256 || (M + S - 1) / S < 4 * kDw2Stage)) S /= 2;
    return S;
}

// ---- round 5: d_w with float32 operands on two fp16 parts (see tri3_kernel) ------------------------------------------------------------
// The contraction index is the row m, so a per-row scale cannot be factored out: the operands are scaled by ONE power of two each, from the
// largest magnitudes of c, g and p (tri_absmax3_kernel: per-workgroup maxima, met in this kernel's prologue) -- A = (g c[:,x]) S_A with
// |g|max |c|max S_A in [2^14, 2^15), B = p S_B.  An element 2^-k below that bound keeps 22 - max(0, k - 17) bits: against the fp32 accumulation
// over 10^4 rows that is nothing.  The split happens between the staging registers and LDS; otherwise tri_dw2_kernel's scheme at half its stage
// depth (six tiles of hi | lo per buffer).
__global__ __launch_bounds__(256) void tri_absmax3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                          size_t n4, float* __restrict__ out) {
    __shared__ float red[4][3];
    float ma = 0.f, mb = 0.f, mc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 va = reinterpret_cast<const f32x4*>(a)[i], vb = reinterpret_cast<const f32x4*>(b)[i], vc = reinterpret_cast<const f32x4*>(c)[i];
        ma = fmaxf(ma, fmaxf(fmaxf(fabsf(va[0]), fabsf(va[1])), fmaxf(fabsf(va[2]), fabsf(va[3]))));
        mb = fmaxf(mb, fmaxf(fmaxf(fabsf(vb[0]), fabsf(vb[1])), fmaxf(fabsf(vb[2]), fabsf(vb[3]))));
        mc = fmaxf(mc, fmaxf(fmaxf(fabsf(vc[0]), fabsf(vc[1])), fmaxf(fabsf(vc[2]), fabsf(vc[3]))));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        ma = fmaxf(ma, __shfl_xor(ma, o));
        mb = fmaxf(mb, __shfl_xor(mb, o));
        mc = fmaxf(mc, __shfl_xor(mc, o));
    }
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6][0] = ma;
        red[threadIdx.x >> 6][1] = mb;
        red[threadIdx.x >> 6][2] = mc;
    }
    __syncthreads();
    if (threadIdx.x < 3)
        out[blockIdx.x * 3 + threadIdx.x] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]), fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

constexpr int kDw3Threads = 1024, kDw3Stage = 32, kDw3XB = 2, kDw3MaxBlocks = 64;
constexpr int kDw3Pitch = 128 * 2 + 32;

__global__ __launch_bounds__(kDw3Threads) void tri_dw3_kernel(const float* __restrict__ c, const float* __restrict__ g, const float* __restrict__ p,
                                                              const float* __restrict__ maxes, int M, int X, int KC, int S,
                                                              float* __restrict__ part) {
    constexpr int HY = 128;
    extern __shared__ __attribute__((aligned(16))) char dw3_smem[];   // two stage buffers of (XB + 1) tiles in hi | lo parts
    constexpr int kTileBytes = kDw3Stage * kDw3Pitch, kBufBytes = 2 * (kDw3XB + 1) * kTileBytes;
    __shared__ float mx[3];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave & 3, wn = wave >> 2;
    if (wave < 3) {   // the three maxima: wave k reduces column k of the [64][3] per-workgroup maxima
        float m = maxes[lane * 3 + wave];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) mx[wave] = m;
    }
    __syncthreads();
    float sA, iA, sB, iB;
    pow2_scale(mx[0] * mx[1], sA, iA);   // (maxes: c, g, p)
    pow2_scale(mx[2], sB, iB);
    const int s = blockIdx.x % S, x0 = (blockIdx.x / S) * kDw3XB;
    const int k_begin = s * KC, k_end = min(M, k_begin + KC), rows = k_end - k_begin;
    // staging: a tile row is 128 float32 = 32 x 16 bytes; 1024 threads cover the 32 rows of a stage in one pass
    const int c4 = tid & 31, r0 = tid >> 5;
    f32x4 rg, rp;
    f32x2 rc;
    auto fetch = [&](int ks) {
        const int k = k_begin + ks + r0;
        const bool ok = k < k_end;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        rg = ok ? *reinterpret_cast<const f32x4*>(g + (size_t)k * HY + c4 * 4) : z;
        rp = ok ? *reinterpret_cast<const f32x4*>(p + (size_t)k * HY + c4 * 4) : z;
        rc = ok ? *reinterpret_cast<const f32x2*>(c + (size_t)k * X + x0) : f32x2{0.f, 0.f};
    };
    auto put = [&](char* tile_hi, int off, const f32x4& v, float f) {   // (v f) -> hi | lo, 4 elements = 8 bytes per part
        uint2 h, l;
        split_f16_pair(v[0] * f, v[1] * f, h.x, l.x);
        split_f16_pair(v[2] * f, v[3] * f, h.y, l.y);
        *reinterpret_cast<uint2*>(tile_hi + off) = h;
        *reinterpret_cast<uint2*>(tile_hi + kTileBytes + off) = l;
    };
    auto stash = [&](int buf) {
        char* base = dw3_smem + buf * kBufBytes;
        const int off = r0 * kDw3Pitch + c4 * 8;
        put(base + 2 * kDw3XB * kTileBytes, off, rp, sB);
#pragma unroll
        for (int xb = 0; xb < kDw3XB; ++xb) put(base + 2 * xb * kTileBytes, off, rg, rc[xb] * sA);
    };
    const int g4 = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int row_off = (4 * g4 + q4) * kDw3Pitch + p4 * 8;
    auto frag = [&](const char* t) { return __builtin_bit_cast(f16x8, tr_frag(tr_read(t), tr_read(t + 16 * kDw3Pitch))); };
    f32x4 acc[kDw3XB][2][2] = {};
    fetch(0);
    stash(0);
    if (kDw3Stage < rows) fetch(kDw3Stage);
    __syncthreads();
    for (int ks = 0, buf = 0; ks < rows; ks += kDw3Stage, buf ^= 1) {
        const char* base = dw3_smem + buf * kBufBytes;
        f16x8 fbh[2], fbl[2];
        const char* pb = base + 2 * kDw3XB * kTileBytes + row_off + (wn * 32) * 2;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fbh[j] = frag(pb + j * 32);
            fbl[j] = frag(pb + kTileBytes + j * 32);
        }
#pragma unroll
        for (int xb = 0; xb < kDw3XB; ++xb) {
            const char* pa = base + 2 * xb * kTileBytes + row_off + (wm * 32) * 2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const f16x8 fah = frag(pa + i * 32), fal = frag(pa + kTileBytes + i * 32);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[xb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fal, fbh[j], acc[xb][i][j], 0, 0, 0);
                    acc[xb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah, fbl[j], acc[xb][i][j], 0, 0, 0);
                    acc[xb][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fah, fbh[j], acc[xb][i][j], 0, 0, 0);
                }
            }
        }
        if (ks + kDw3Stage < rows) {
            stash(buf ^ 1);
            if (ks + 2 * kDw3Stage < rows) fetch(ks + 2 * kDw3Stage);
        }
        __syncthreads();
    }
    const float un = iA * iB;
#pragma unroll
    for (int xb = 0; xb < kDw3XB; ++xb) {
        if (x0 + xb >= X) break;
        float* out = part + ((size_t)s * X + x0 + xb) * HY * HY;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    out[(size_t)(wm * 32 + i * 16 + 4 * g4 + r) * HY + wn * 32 + j * 16 + (lane & 15)] = acc[xb][i][j][r] * un;
    }
}

static bool dw3_applies(int M, int X, int H, int Y) { return H == 128 && Y == 128 && X == 128 && M >= 1024; }   // (X = 128: the three operands share a shape)
static int dw3_splits(int M, int X) {
    int S = 8;
    while (S > 1 && ((X / kDw3XB) * (S / 2) >= 256 || (M + S - 1) / S < 8 * kDw3Stage)) S /= 2;
    return S;
}

// the float32 backward on fp16 parts (X = H = Y = 128): scratch behind the permuted float32 weight copies
struct Tri3BwdPlan {
    size_t off_wA, off_wB, off_g, off_max, off_part, bytes;
    Tri3BwdPlan(int M, int X, int H, int Y, size_t base) {
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        off_wA = base;
        off_wB = off_wA + tri3_carve(nullptr, (size_t)Y * X, nullptr);
        off_g = off_wB + tri3_carve(nullptr, (size_t)X * Y, nullptr);
        off_max = off_g + tri3_carve(nullptr, (size_t)M, nullptr);
        off_part = off_max + up(sizeof(float) * 3 * kDw3MaxBlocks);
        size_t pb = sizeof(float) * (size_t)dw3_splits(M, X) * X * H * Y;
        pb = pb > tri3_part_bytes(M, Y) ? pb : tri3_part_bytes(M, Y);
        pb = pb > tri3_part_bytes(M, X) ? pb : tri3_part_bytes(M, X);
        bytes = off_part + up(pb);
    }
};

struct TriBwdPlan {   // scratch carving (bytes), shared by the size query and the launcher
    size_t esz, Mp, off_wA, off_wB, off_gB, off_cT, off_gT, off_pT, off_part, bytes;
    TriBwdPlan(int M, int X, int H, int Y, bool f32in) {
        esz = f32in ? 4 : 2;
        const int kw = f32in ? 16 : 32;
        Mp = ((size_t)M + kw - 1) / kw * kw;
        auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
        const size_t wbytes = up((size_t)X * H * Y * esz);
        off_wA = 0;
        off_wB = wbytes;
        off_gB = 2 * wbytes;
        off_cT = off_gB + up((size_t)M * H * esz);
        off_gT = off_cT + up((size_t)X * Mp * esz);
        off_pT = off_gT + up((size_t)H * Mp * esz);
        off_part = off_pT + up((size_t)Y * Mp * esz);
        size_t pb = dw2_applies(M, X, H, Y, f32in) ? sizeof(float) * (size_t)dw2_splits(M, X) * X * H * Y : 0;
        // the slabs of tri2_kernel's partial sums share the region (the three launches are stream-ordered); roles permute (X, H, Y)
        if (tri2_applies(M, Y, X, H, f32in)) pb = pb > tri2_part_bytes(M, Y) ? pb : tri2_part_bytes(M, Y);
        if (tri2_applies(M, X, Y, H, f32in)) pb = pb > tri2_part_bytes(M, X) ? pb : tri2_part_bytes(M, X);
        bytes = off_part + up(pb);
        if (f32in && tri3_applies(M, X, H, Y) && dw3_applies(M, X, H, Y)) {   // the fp16-parts path: only the permuted weight copies of the layout above
            const size_t b3 = Tri3BwdPlan(M, X, H, Y, off_gB).bytes;
            bytes = bytes > b3 ? bytes : b3;
        }
    }
};

// float32 operands, X = H = Y = 128: every product on two fp16 parts per operand (tri3_kernel, tri_dw3_kernel)
static int run_tri_backward_f16x3(const float* child, const float* w, const float* parent, const float* g, int M, int X, int H, int Y, char* ws,
                                  const TriBwdPlan& p, float* d_child, float* d_w, float* d_parent, hipStream_t s) {
    const Tri3BwdPlan q(M, X, H, Y, p.off_gB);
    float *wA = (float*)(ws + p.off_wA), *wB = (float*)(ws + p.off_wB);
    float* part = reinterpret_cast<float*>(ws + q.off_part);
    Tri3Ops wAo, wBo, go;
    tri3_carve(ws + q.off_wA, (size_t)Y * X, &wAo);
    tri3_carve(ws + q.off_wB, (size_t)X * Y, &wBo);
    tri3_carve(ws + q.off_g, (size_t)M, &go);
    if (d_child || d_parent) {
        hipLaunchKernelGGL((tri_permute2_kernel<float>), dim3(X, (Y + 31) / 32), dim3(256), 0, s, w, wA, wB, X, H, Y);
        launch_split_rows(g, (size_t)M, go, s);
    }
    if (d_child) {   // roles (c, w, p) := (parent, wA [Y][X][H], g)
        launch_split_rows(wA, (size_t)Y * X, wAo, s);
        if (int rc = launch_tri3(parent, wAo, go, M, Y, d_child, part, s)) return rc;
    }
    if (d_parent) {  // roles (c, w, p) := (child, wB [X][Y][H], g)
        launch_split_rows(wB, (size_t)X * Y, wBo, s);
        if (int rc = launch_tri3(child, wBo, go, M, X, d_parent, part, s)) return rc;
    }
    if (d_w) {
        float* maxes = reinterpret_cast<float*>(ws + q.off_max);
        hipLaunchKernelGGL(tri_absmax3_kernel, dim3(kDw3MaxBlocks), dim3(256), 0, s, child, g, parent, (size_t)M * 128 / 4, maxes);
        const int S = dw3_splits(M, X);
        const int KC = ((M + S - 1) / S + kDw3Stage - 1) / kDw3Stage * kDw3Stage;
        const size_t lds = 2 * (size_t)2 * (kDw3XB + 1) * kDw3Stage * kDw3Pitch;   // 108 KB
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_dw3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(tri_dw3_kernel, dim3((X / kDw3XB) * S), dim3(kDw3Threads), lds, s, child, g, parent, maxes, M, X, KC, S, part);
        const size_t n = (size_t)X * H * Y;
        hipLaunchKernelGGL(tri_dw2_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, part, S, n, d_w);
    }
    return check_launch("trilinear_backward (fp16 parts)");
}

template <bool F32IN>
static int run_tri_backward(const void* child, const void* w, const void* parent, const void* g, bool g_is_bf16, int M, int X, int H,
                            int Y, char* ws, const TriBwdPlan& p, float* d_child, float* d_w, float* d_parent, hipStream_t s) {
    using T = typename MfmaCfg<F32IN>::T;
    if constexpr (F32IN) {
        if (tri3_applies(M, X, H, Y) && dw3_applies(M, X, H, Y) && !VLG_ENV("VLG_TRI_F32_EXACT"))
            return run_tri_backward_f16x3((const float*)child, (const float*)w, (const float*)parent, (const float*)g, M, X, H, Y, ws, p, d_child, d_w,
                                          d_parent, s);
    }
    T *wA = (T*)(ws + p.off_wA), *wB = (T*)(ws + p.off_wB), *gB = (T*)(ws + p.off_gB);
    T *cT = (T*)(ws + p.off_cT), *gT = (T*)(ws + p.off_gT), *pT = (T*)(ws + p.off_pT);
    const int pblocks = 1024;
    const void* g_op = g;   // the cotangent in the operand type, row-major
    if constexpr (!F32IN) {
        if (g_is_bf16) gB = (T*)const_cast<void*>(g);   // already in the operand type: no copy
        else hipLaunchKernelGGL(tri_cast_bf16_kernel, dim3(512), dim3(256), 0, s, (const float*)g, (uint16_t*)gB, (size_t)M * H);
        g_op = gB;
    }
    bool permuted = false;
    if (d_child && d_parent && H <= 128 && X % 16 == 0 && X <= 16 * kTriWaves * kTriHPW && Y % 16 == 0 && Y <= 16 * kTriWaves * kTriHPW) {
        hipLaunchKernelGGL((tri_permute2_kernel<T>), dim3(X, (Y + 31) / 32), dim3(256), 0, s, (const T*)w, wA, wB, X, H, Y);
        permuted = true;
    }
    if (d_child) {   // d_child[m,x] = sum_y p[m,y] * sum_h wA[y,x,h] g[m,h]: roles (c, w, p) := (p, wA, g); "X" = Y, "H" = X, "Y" = H
        if (X % 16 || X > 16 * kTriWaves * kTriHPW) return set_error(VLG_ERR_SHAPE, "trilinear_backward: X=%d must be a multiple of 16 and <= 128", X);
        if (!permuted) hipLaunchKernelGGL((tri_permute_kernel<T>), dim3(pblocks), dim3(256), 0, s, (const T*)w, wA, X, H, Y, 0);
        if (int rc = dispatch_tri(parent, wA, g_op, M, Y, X, H, F32IN, d_child, s, F32IN ? nullptr : reinterpret_cast<float*>(ws + p.off_part))) return rc;
    }
    if (d_parent) {  // d_parent[m,y] = sum_x c[m,x] * sum_h wB[x,y,h] g[m,h]: roles (c, w, p) := (c, wB, g); "H" = Y, "Y" = H
        if (Y % 16 || Y > 16 * kTriWaves * kTriHPW) return set_error(VLG_ERR_SHAPE, "trilinear_backward: Y=%d must be a multiple of 16 and <= 128", Y);
        if (!permuted) hipLaunchKernelGGL((tri_permute_kernel<T>), dim3(pblocks), dim3(256), 0, s, (const T*)w, wB, X, H, Y, 1);
        if (int rc = dispatch_tri(child, wB, g_op, M, X, Y, H, F32IN, d_parent, s, F32IN ? nullptr : reinterpret_cast<float*>(ws + p.off_part))) return rc;
    }
    if (d_w && dw2_applies(M, X, H, Y, F32IN)) {
        if constexpr (!F32IN) {
            const int S = dw2_splits(M, X);
            const int KC = ((M + S - 1) / S + kDw2Stage - 1) / kDw2Stage * kDw2Stage;
            const int Sx = S;   // a trailing chunk may be empty (its partial tile is then zero)
            float* part = reinterpret_cast<float*>(ws + p.off_part);
            const size_t lds = 2 * (size_t)(kDw2XB + 1) * kDw2Stage * kDw2Pitch;   // 110 KB
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tri_dw2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            hipLaunchKernelGGL(tri_dw2_kernel, dim3((X / kDw2XB) * Sx), dim3(kDw2Threads), lds, s, (const uint16_t*)child, (const uint16_t*)gB,
                               (const uint16_t*)parent, M, X, KC, Sx, part);
            const size_t n = (size_t)X * H * Y;
            hipLaunchKernelGGL(tri_dw2_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, part, Sx, n, d_w);
        }
    } else if (d_w) {
        if (Y > 16 * kDwYT || Y % 16) return set_error(VLG_ERR_SHAPE, "trilinear_backward: Y=%d must be a multiple of 16 and <= 128", Y);
        const int Mp = (int)p.Mp;
        dim3 tb(256);
        hipLaunchKernelGGL((tri_transpose_kernel<T, false>), dim3((Mp + 31) / 32, (X + 31) / 32), tb, 0, s, child, cT, M, X, Mp);
        if (!F32IN && g_is_bf16)
            hipLaunchKernelGGL((tri_transpose_kernel<T, false>), dim3((Mp + 31) / 32, (H + 31) / 32), tb, 0, s, g, gT, M, H, Mp);
        else
            hipLaunchKernelGGL((tri_transpose_kernel<T, true>), dim3((Mp + 31) / 32, (H + 31) / 32), tb, 0, s, g, gT, M, H, Mp);
        hipLaunchKernelGGL((tri_transpose_kernel<T, false>), dim3((Mp + 31) / 32, (Y + 31) / 32), tb, 0, s, parent, pT, M, Y, Mp);
        const size_t lds = sizeof(float) * 4 * (size_t)(kDwSplit - 1) * kDwHT * kDwYT * 64;   // 96 KB
        auto k = tri_dw_kernel<F32IN>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        hipLaunchKernelGGL(k, dim3(X, ((H >> 4) + kDwHT - 1) / kDwHT), dim3(64 * kDwSplit), lds, s, cT, gT, pT, Mp, X, H, Y, d_w);
    }
    return check_launch("trilinear_backward");
}

static int check_dims(const char* what, int M, int X, int H, int Y) {
    if (M < 0 || X < 1 || H < 1 || Y < 1) return set_error(VLG_ERR_SHAPE, "%s: bad shape M=%d X=%d H=%d Y=%d", what, M, X, H, Y);
    if (H % 16 || H > 16 * kTriWaves * kTriHPW) return set_error(VLG_ERR_SHAPE, "%s: H=%d must be a multiple of 16 and <= 128", what, H);
    if (X > 256) return set_error(VLG_ERR_SHAPE, "%s: X=%d > 256", what, X);
    return 0;
}

}  // namespace vlg

extern "C" {

int vlg_trilinear(const void* child, const void* w, const void* parent, int M, int X, int H, int Y, int in_dtype, float* out,
                  void* stream) {
    using namespace vlg;
    if (int rc = check_dims("trilinear", M, X, H, Y)) return rc;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "trilinear: in_dtype %d", in_dtype);
    if (M == 0) return 0;
    if (!child || !w || !parent || !out) return set_error(VLG_ERR_ARG, "trilinear: null buffer");
    return dispatch_tri(child, w, parent, M, X, H, Y, in_dtype == VLG_F32, out, (hipStream_t)stream);
}

size_t vlg_trilinear_workspace(int M, int X, int H, int Y, int in_dtype) {
    if (M < 1 || X < 1 || H < 1 || Y < 1) return 0;
    if (in_dtype == VLG_F32) return vlg::tri3_applies(M, X, H, Y) ? vlg::tri3_fwd_bytes(M, X) : 0;
    return vlg::tri2_applies(M, X, H, Y, false) ? vlg::tri2_part_bytes(M, X) : 0;
}

int vlg_trilinear_ws(const void* child, const void* w, const void* parent, int M, int X, int H, int Y, int in_dtype, void* ws,
                     size_t ws_bytes, float* out, void* stream) {
    using namespace vlg;
    if (int rc = check_dims("trilinear", M, X, H, Y)) return rc;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "trilinear: in_dtype %d", in_dtype);
    if (M == 0) return 0;
    if (!child || !w || !parent || !out) return set_error(VLG_ERR_ARG, "trilinear: null buffer");
    const size_t need = vlg_trilinear_workspace(M, X, H, Y, in_dtype);
    float* part = (need && ws && ws_bytes >= need) ? (float*)ws : nullptr;   // without it: the two-range form
    return dispatch_tri(child, w, parent, M, X, H, Y, in_dtype == VLG_F32, out, (hipStream_t)stream, part);
}

size_t vlg_trilinear_backward_workspace(int M, int X, int H, int Y, int in_dtype) {
    if (M < 1 || X < 1 || H < 1 || Y < 1) return 0;
    return vlg::TriBwdPlan(M, X, H, Y, in_dtype == VLG_F32).bytes;
}

int vlg_trilinear_backward(const void* child, const void* w, const void* parent, const float* g, int M, int X, int H, int Y,
                           int in_dtype, void* ws, size_t ws_bytes, float* d_child, float* d_w, float* d_parent, void* stream) {
    return vlg_trilinear_backward_g(child, w, parent, g, VLG_F32, M, X, H, Y, in_dtype, ws, ws_bytes, d_child, d_w, d_parent, stream);
}

int vlg_trilinear_backward_g(const void* child, const void* w, const void* parent, const void* g, int g_dtype, int M, int X, int H,
                             int Y, int in_dtype, void* ws, size_t ws_bytes, float* d_child, float* d_w, float* d_parent, void* stream) {
    using namespace vlg;
    if (int rc = check_dims("trilinear_backward", M, X, H, Y)) return rc;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "trilinear_backward: in_dtype %d", in_dtype);
    if (g_dtype != VLG_F32 && !(g_dtype == VLG_BF16 && in_dtype == VLG_BF16))
        return set_error(VLG_ERR_DTYPE, "trilinear_backward: a bf16 cotangent needs bf16 features (g_dtype %d, in_dtype %d)", g_dtype, in_dtype);
    if (Y != 32 && Y != 64 && Y != 128) return set_error(VLG_ERR_SHAPE, "trilinear_backward: Y=%d (supported: 32, 64, 128)", Y);
    if (H != 32 && H != 64 && H != 128) return set_error(VLG_ERR_SHAPE, "trilinear_backward: H=%d (supported: 32, 64, 128)", H);
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) {
        if (d_w) {
            hipError_t e = hipMemsetAsync(d_w, 0, sizeof(float) * (size_t)X * H * Y, s);
            if (e != hipSuccess) return set_error((int)e, "trilinear_backward: %s", hipGetErrorString(e));
        }
        return 0;
    }
    if (!child || !w || !parent || !g) return set_error(VLG_ERR_ARG, "trilinear_backward: null buffer");
    const TriBwdPlan p(M, X, H, Y, in_dtype == VLG_F32);
    if (!ws || ws_bytes < p.bytes) return set_error(VLG_ERR_WORKSPACE, "trilinear_backward: workspace %zu bytes < %zu", ws_bytes, p.bytes);
    const bool gb = g_dtype == VLG_BF16;
    return in_dtype == VLG_F32 ? run_tri_backward<true>(child, w, parent, g, gb, M, X, H, Y, (char*)ws, p, d_child, d_w, d_parent, s)
                               : run_tri_backward<false>(child, w, parent, g, gb, M, X, H, Y, (char*)ws, p, d_child, d_w, d_parent, s);
}

}  // extern "C"
