// vlg_attn.hip -- the attention-fuse feeding the parser (gfx950) and its C-ABI entry points.
//
//   vlg_attn_fuse           : DependencyBoxRel._forward, src/model/joint.py:670-674
//   vlg_attn_fuse_backward  : its adjoint (what autograd derives for those lines), for training through the fuse
//
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_dp_core.h"   // F32In / BF16In element loaders
#include "vlg_mfma.h"      // bf16x8 fragments

namespace vlg {

constexpr int kAlignThreads = 256;   // generic kernel block size
template <typename In>
constexpr bool kIsBF16 = sizeof(typename In::T) == 2;
typedef short short8 __attribute__((ext_vector_type(8)));

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

// ---- bf16 features: the same chains on v_mfma_f32_16x16x32_bf16 (16x the fp32 matrix rate) --------------------------------
// Products of bf16 values are exact in fp32 and the accumulation is fp32, so GEMM 1 loses nothing against the widened-fp32 form.
// A K step is 32 wide: lane (r, g) holds k = 8g .. 8g+7.  Operands that are rows in memory (vis rows for GEMM 1, vis_mid rows for
// GEMM 3) are ONE 16-byte read per fragment.  Accumulator tiles as the next product's B operand: two tiles (t0, t1) = (2s, 2s+1) make
// one K step -- element j < 4 is register j of tile t0 (key 16 t0 + 4g + j), element j >= 4 register j-4 of tile t1 -- and the A
// fragment gathers the same keys (single 16-bit reads, as the fp32 form does).  Probabilities enter GEMM 2 as bf16 hi + lo pairs
// (two MFMAs, 2^-17 relative); the adjoint's cotangent operands (dy, dS, P) as single bf16 values -- what a bf16 training step
// carries between its GEMMs anyway.
__device__ __forceinline__ bf16x8 frag_zero() { return __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u)); }
__device__ __forceinline__ bf16x8 frag_ld16(const uint16_t* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); }
__device__ __forceinline__ bf16x8 frag_ld8x2(const uint16_t* p0, const uint16_t* p1) {
    const uint2 a = *reinterpret_cast<const uint2*>(p0), b = *reinterpret_cast<const uint2*>(p1);
    return __builtin_bit_cast(bf16x8, make_uint4(a.x, a.y, b.x, b.y));
}
__device__ __forceinline__ unsigned short buf_ld16(__amdgpu_buffer_rsrc_t r, int lane_off, int uni_off) {
    return (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, lane_off * 2, uni_off * 2, 0);
}
// fragment of a K step from two accumulator tiles' worth of gathered 16-bit values (second tile absent: zeros)
__device__ __forceinline__ bf16x8 frag_pack(const unsigned short (&t0)[4], const unsigned short (&t1)[4], bool has1) {
    short8 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = (short)t0[j];
        v[4 + j] = has1 ? (short)t1[j] : (short)0;
    }
    return __builtin_bit_cast(bf16x8, v);
}

// Products whose A operand is the TRANSPOSE of rows in memory (GEMM 2: vis_mid^T, GEMM 4: vis^T; the contraction index -- the key --
// is the slow dimension): the rows of 32 keys x 64 columns are staged through a wave-private LDS image (16-byte reads and writes)
// and come back as fragments through ds_read_b64_tr_b16.  Gathering the same fragments as single 16-bit reads costs one
// vector-memory instruction per 2 bytes per lane -- 256 per 64-key step against 32 here -- and the texture-address unit, not the
// matrix pipe, then sets the pace (measured: 9.4 us per 64-key step either way round, fp32 or bf16 MFMAs).
//   acc[ct] += src^T[16ct + row][key] . B[key][word]   over the keys of the step at v0 (T region tiles; rows clamped to vmax - 1:
//   their B entries are zero), ct < NT; Bh / Bl: the B fragments of each 32-key half (hi and, LO, lo parts).
// Image: [32 keys][80] bf16 (pitch 160 B: the 8 rows a half-wave reads sit on 8 distinct bank groups), two buffers, stages
// (half, group of four output tiles) software-pipelined two deep in registers.  One wave = one workgroup: no barrier, the LDS
// executes a wave's operations in order.
constexpr int kStagePitch = 160;                       // bytes per image row
constexpr int kStageBytes = 2 * 32 * kStagePitch;      // two buffers

template <int T, int NT, bool LO>
__device__ __forceinline__ void attn_keys_product_lds(f32x4 (&acc)[NT], const uint16_t* __restrict__ src, int ncols, int v0, int vmax,
                                                      const bf16x8 (&Bh)[(T + 1) / 2], const bf16x8 (&Bl)[(T + 1) / 2], char* lds) {
    constexpr int NS = (T + 1) / 2, NG = NT / 4, NSTG = NS * NG;
    const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
    const int lrow = lane >> 3, lcol = 8 * (lane & 7);   // staging: 8 lanes x 16 bytes per 64-column row, 8 rows per instruction
    uint4 regs[2][4];
    auto fetch = [&](int stg, uint4 (&rg)[4]) {
        const int hf = stg / NG, cg = stg % NG;
        const int col = min(64 * cg + lcol, ncols - 8);   // (columns past the row: tiles that are never used)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int key = min(v0 + 32 * hf + 8 * i + lrow, vmax - 1);
            rg[i] = *reinterpret_cast<const uint4*>(src + (size_t)key * ncols + col);
        }
    };
    fetch(0, regs[0]);
    if (NSTG > 1) fetch(1, regs[1]);
    const char* rd = lds + (4 * g + q) * kStagePitch + p4 * 8;
#pragma unroll
    for (int stg = 0; stg < NSTG; ++stg) {
        char* buf = lds + (stg & 1) * (32 * kStagePitch);
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(buf + (8 * i + lrow) * kStagePitch + lcol * 2) = regs[stg & 1][i];
        if (stg + 2 < NSTG) fetch(stg + 2, regs[stg & 1]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is written (and the compiler keeps the reads below it)
        const int hf = stg / NG, cg = stg % NG;
        const char* rb = rd + (stg & 1) * (32 * kStagePitch);
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const bf16x8 a = tr_frag(tr_read(rb + c4 * 32), tr_read(rb + 16 * kStagePitch + c4 * 32));
            acc[4 * cg + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Bh[hf], acc[4 * cg + c4], 0, 0, 0);
            if constexpr (LO) acc[4 * cg + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, Bl[hf], acc[4 * cg + c4], 0, 0, 0);
        }
        asm volatile("" ::: "memory");   // (the next stage's writes stay behind these reads)
    }
}

// GEMM 1 of one region chunk: S^T[region v0+16t+4g+n][word r] = <vis[region], txt[word]>  (joint.py:670-672).
// trow = this lane's word row (element 0).
template <typename In, int T, bool F32MATH = false>   // F32MATH: bf16 features widened on load, fp32 MFMAs (the forward of a few dozen keys)
__device__ __forceinline__ void attn_scores(f32x4 (&S)[T], const typename In::T* trow, const typename In::T* vis_b, int V,
                                            int d, int v0, int r, int g) {
#pragma unroll
    for (int t = 0; t < T; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const typename In::T* vrow[T];
#pragma unroll
    for (int t = 0; t < T; ++t) vrow[t] = vis_b + (size_t)min(v0 + 16 * t + r, V - 1) * d;
    if constexpr (kIsBF16<In> && !F32MATH) {
        constexpr int KJ = 4;   // 32-feature steps per chunk (128 features)
        for (int k0 = 0; k0 < d; k0 += 32 * KJ) {
            int koff[KJ];
            bool kin[KJ];      // d is a multiple of 16, not of 32: the last step may hold 16 features (lanes g >= 2 contribute zeros)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int k = k0 + 32 * j + 8 * g;
                kin[j] = k < d;
                koff[j] = kin[j] ? k : 0;
            }
            bf16x8 wf[KJ];
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const bf16x8 x = frag_ld16(trow + koff[j]);
                wf[j] = kin[j] ? x : frag_zero();
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bf16x8 rf[KJ];
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const bf16x8 x = frag_ld16(vrow[t] + koff[j]);
                    rf[j] = kin[j] ? x : frag_zero();
                }
#pragma unroll
                for (int j = 0; j < KJ; ++j) S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rf[j], wf[j], S[t], 0, 0, 0);
            }
        }
    } else {
        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
        trow += 4 * g;
#pragma unroll
        for (int t = 0; t < T; ++t) vrow[t] += 4 * g;
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
}

// ---- bf16 features: the forward tile as ONE stream of loads running a whole 64-key step ahead of the arithmetic ---------------
// Counters of the first key-split build (1152 wavefronts, ~1 per SIMD): 73 % of the wave cycles in s_waitcnt -- every step began by
// requesting its key rows and waited out the full memory latency, eight more times per step for the staged vis_mid pieces; the
// matrix pipe was busy 7 % of the time.  Nothing a step loads depends on what the previous one computed, so the loads of step k+1
// -- the vis fragments of GEMM 1 (T x 4 x 16 bytes per lane) and the eight (half, channel group) pieces of the vis_mid tile
// (8 x 4 x 16 bytes per lane) -- are issued during step k into the registers step k has just drained: ~200 registers of a wave that
// has the SIMD's whole file to itself (the chunk plan keeps to one wave per SIMD).  Past the last step the same addresses clamp to
// the last key row (L1 hits, never used): the loop body issues a fixed number of loads, which keeps the compiler's counted waits exact.
// Addressing: buffer loads -- per-lane byte offsets that never change (one register each), the step / piece offset in a scalar
// register, rows past the sentence's last key read as zeros (no clamps, no per-load address arithmetic: the first build spent a third of
// its vector instructions on 64-bit addresses).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 buf_ld128(__amdgpu_buffer_rsrc_t r, int lane_bytes, int uni_bytes) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, uni_bytes, 0);
}

template <int T>
__device__ __forceinline__ void attn_forward_tile_bf16(f32x4 (&Y)[kAttnMaxCT], float& m_run, float& z_run, const uint16_t* trow,
                                                       const uint16_t* vis_b, const uint16_t* mid_p, int Vall, int d, int h, int r, int g,
                                                       int v_begin, int v_end, bool normalize, char* lds) {
    constexpr int NS = (T + 1) / 2, NG = kAttnMaxCT / 4, NSTG = NS * NG, KJ = 4;
    const int lane = threadIdx.x, q = (lane >> 2) & 3, p4 = lane & 3;
    const int lrow = lane >> 3, lcol = 8 * (lane & 7);   // staging: 8 lanes x 16 bytes per 64-column row, 8 rows per instruction
    const int V = v_end;
    const __amdgpu_buffer_rsrc_t vis_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vis_b, 0, Vall * d * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t mid_rs = __builtin_amdgcn_make_buffer_rsrc((void*)mid_p, 0, Vall * h * 2, 0x00020000);
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run = neg_infinity();
    z_run = 0.f;

    bf16x8 wf0[KJ];   // this lane's word, first 128 features, resident for the whole tile; lanes past a 16-feature tail hold zeros
                      // (the key rows' fragments then need no masking: what they read there is finite data of the next row)
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        const int k = 32 * j + 8 * g;
        const bf16x8 x = frag_ld16(trow + (k < d ? k : 0));
        wf0[j] = k < d ? x : frag_zero();
    }
    u32x4 vf[T][KJ];      // the key rows' fragments of the NEXT step's GEMM 1
    u32x4 ring[NSTG][4];  // the NEXT step's vis_mid pieces
    const int vlane = (r * d + 8 * g) * 2;       // byte offset of this lane's fragment inside a 16-row tile of vis
    const int mlane = (lrow * h + lcol) * 2;     // ... of this lane's 16 bytes inside an 8-row group of vis_mid
    auto fetch_scores = [&](int v0) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < KJ; ++j) vf[t][j] = buf_ld128(vis_rs, vlane + 64 * j, (v0 + 16 * t) * d * 2);
    };
    auto fetch_piece = [&](int v0, int stg, int i) -> u32x4 {   // rows 8i + lrow of piece (half, channel group) = stg
        const int hf = stg / NG, cg = stg % NG;
        return buf_ld128(mid_rs, mlane, ((v0 + 32 * hf + 8 * i) * h + 64 * cg) * 2);
    };
    // Issue order = the order the loop keeps (key rows of GEMM 1 first, then the pieces in stage order): the counted waits at the loop
    // head are the merge of this path and the back edge -- with the key rows requested LAST here the head became s_waitcnt vmcnt(0),
    // i.e. every step waited for the whole next tile it had just requested.
    // (a compiler memory barrier pins the order at the IR level, where loads otherwise sink towards their first use)
    fetch_scores(v_begin);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int stg = 0; stg < NSTG; ++stg) {
#pragma unroll
        for (int i = 0; i < 4; ++i) ring[stg][i] = fetch_piece(v_begin, stg, i);
        asm volatile("" ::: "memory");
    }
    const char* rd = lds + (4 * g + q) * kStagePitch + p4 * 8;
    char* wr = lds + lrow * kStagePitch + lcol * 2;

    for (int v0 = v_begin; v0 < V; v0 += 16 * T) {
        // ---- GEMM 1: S^T[key][word] (joint.py:670-672) ----
        f32x4 S[T];
#pragma unroll
        for (int t = 0; t < T; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KJ; ++j)   // (K step outermost: the T accumulation chains interleave instead of four dependent MFMAs in a row)
#pragma unroll
            for (int t = 0; t < T; ++t)
                S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf[t][j]), wf0[j], S[t], 0, 0, 0);
        for (int k0 = 32 * KJ; k0 < d; k0 += 32 * KJ) {   // features past 128: read in place
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int k = k0 + 32 * j + 8 * g;
                const bool in = k < d;
                const bf16x8 w = frag_ld16(trow + (in ? k : 0));
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const bf16x8 a = frag_ld16(vis_b + (size_t)min(v0 + 16 * t + r, V - 1) * d + (in ? k : 0));
                    S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, in ? w : frag_zero(), S[t], 0, 0, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        fetch_scores(v0 + 16 * T);   // the next step's key rows (zeros past the sentence's end)
        asm volatile("" ::: "memory");
        // ---- softmax over the keys (NO region masking: faithful to joint.py:670-672; only the tile padding is dropped) ----
        float m = m_run;
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                if (v0 + 16 * t + 4 * g + n >= V) S[t][n] = neg_infinity();
                m = fmaxf(m, S[t][n]);
            }
        m = group_max4(m);   // finite: every step holds at least one real key
        const float rescale = __expf(m_run - m);   // 0 on the first step
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
        if (v0 > v_begin) {
#pragma unroll
            for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= rescale;
        }
        bf16x8 Ph[NS], Pl[NS];   // probabilities as bf16 hi + lo: the B fragments of the two 32-key halves
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int t = 2 * s2 + (j >> 2);
                const float p = t < T ? S[t < T ? t : 0][j & 3] : 0.f;
                const __bf16 hi = (__bf16)p;
                Ph[s2][j] = hi;
                Pl[s2][j] = (__bf16)(p - (float)hi);
            }
        // ---- GEMM 2: Y^T[channel][word] += vis_mid^T[channel][key] . P[key][word], through the transposition image ----
#pragma unroll
        for (int stg = 0; stg < NSTG; ++stg) {
            const int bo = (stg & 1) * (32 * kStagePitch);
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4*>(wr + bo + 8 * i * kStagePitch) = ring[stg][i];
#pragma unroll
            for (int i = 0; i < 4; ++i) ring[stg][i] = fetch_piece(v0 + 16 * T, stg, i);   // the same piece of the next step
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the image is written (and the compiler keeps the reads below it)
            const int hf = stg / NG, cg = stg % NG;
            bf16x8 a[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) a[c4] = tr_frag(tr_read(rd + bo + c4 * 32), tr_read(rd + bo + 16 * kStagePitch + c4 * 32));
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) Y[4 * cg + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c4], Ph[hf], Y[4 * cg + c4], 0, 0, 0);
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) Y[4 * cg + c4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c4], Pl[hf], Y[4 * cg + c4], 0, 0, 0);
            asm volatile("" ::: "memory");   // (the next stage's writes stay behind these reads)
        }
    }
    if (normalize) {
        const float zinv = 1.f / z_run;
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= zinv;
    }
}

// The whole forward of one 16-word tile up to (not including) the residual: on return Y^T[channel 16ct+4g+n][word r]
// holds softmax_v(S) . vis_mid and (m_run, z_run) the softmax statistics of word r.
// Regions stream through in chunks of 16 T (one chunk when V <= 64, the benchmark's case); between chunks the
// accumulators are rescaled by exp(old max - new max), the usual streaming softmax.
// Keys [v_begin, v_end) of the sentence's V (v_begin a multiple of 16 T, v_begin < v_end <= V): the whole sentence in the one-pass
// kernels, one key chunk in the key-split kernels (which pass normalize = false and combine the chunks' (max, sum, Y) afterwards).
template <typename In, int T>
__device__ __forceinline__ void attn_forward_tile(f32x4 (&Y)[kAttnMaxCT], float& m_run, float& z_run,
                                                  const typename In::T* trow, const typename In::T* vis_b,
                                                  const typename In::T* mid_ptr, int V, int d, int h, int r, int g,
                                                  int v_begin, int v_end, bool normalize, char* lds) {
    // bf16 features: the streaming bf16-MFMA tile for key ranges of several 64-key steps (T = 4: V > 48); a few dozen keys (object-only
    // layouts, one step) keep round 3's form -- values widened on load, exact fp32 MFMAs, vis_mid gathered straight into the operand ring:
    // with nothing to stream, the LDS image and the step-ahead loads only add latency (measured at V = 36, B = 256: 24.5 vs 18.6 us)
    if constexpr (kIsBF16<In> && T == 4) {
        attn_forward_tile_bf16<T>(Y, m_run, z_run, trow, vis_b, mid_ptr, V, d, h, r, g, v_begin, v_end, normalize, lds);
        return;
    }
    const int CT = h >> 4;
    const __amdgpu_buffer_rsrc_t mid_b =
        __builtin_amdgcn_make_buffer_rsrc((void*)mid_ptr, 0, (int)(V * h * sizeof(typename In::T)), 0x00020000);
    V = v_end;   // everything below clamps and masks against the end of this key range
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    m_run = neg_infinity();
    z_run = 0.f;
    for (int v0 = v_begin; v0 < V; v0 += 16 * T) {
        f32x4 S[T];
        attn_scores<In, T, true>(S, trow, vis_b, V, d, v0, r, g);
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
        if (v0 > v_begin) {
#pragma unroll
            for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= rescale;
        }

        // ---- GEMM 2: Y^T[channel 16ct+4g+n][word r] += mid[region][channel] * exp(score - max) ----
        {
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
    }
    if (normalize) {
        const float zinv = 1.f / z_run;
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= zinv;
    }
}

// ---- residual + LayerNorm over channels (biased variance like nn.LayerNorm), shared by the one-pass and the combine kernels ----
// Y arrives in accumulator layout (lane (r,g), register (ct,n) <-> word q0 + r, channel 16ct + 4g + n).  The enc_x rows were
// requested by the caller as whole rows (one instruction = one contiguous row) before its long phase; they meet the accumulators
// in LDS (tile: [16 words][h + 4] floats), and the result leaves as whole rows again.  Reading / writing in accumulator layout
// directly moves 64-byte pieces.
template <typename In>
__device__ __forceinline__ void attn_load_residual_rows(float4 (&erows)[16], const typename In::T* enc_x, int b, int q0, int Lq, int h,
                                                        int lane) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
        erows[i] = ld4(enc_x + ((size_t)b * Lq + min(q0 + i, Lq - 1)) * h + min(4 * lane, h - 4));
}

__device__ __forceinline__ void attn_residual_layernorm_store(f32x4 (&Y)[kAttnMaxCT], const float4 (&erows)[16], float* tile,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              int b, int q0, int Lq, int h, float eps, float* __restrict__ out) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int CT = h >> 4, hp = h + 4;
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
__global__ __launch_bounds__(64) void attn_fuse_mfma_kernel(
    const typename In::T* __restrict__ vis, const typename In::T* __restrict__ txt,
    const typename In::T* __restrict__ vis_mid, const typename In::T* __restrict__ enc_x,
    const float* __restrict__ gamma, const float* __restrict__ beta, int Lq, int V, int d, int h, float eps,
    float* __restrict__ out) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 16;
    const int qw = min(q0 + r, Lq - 1);   // this lane's word (clamped; rows past Lq are never stored)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw + kStageBytes);   // [16 words][h + 4], behind the transposition image
    float4 erows[16];   // residual rows, needed only after both GEMMs: the read latency is free
    attn_load_residual_rows<In>(erows, enc_x, b, q0, Lq, h, lane);

    f32x4 Y[kAttnMaxCT];
    float m_run, z_run;
    attn_forward_tile<In, T>(Y, m_run, z_run, txt + ((size_t)b * (Lq + 1) + 1 + qw) * d,   // root slot skipped: txt[:, 1:]
                             vis + (size_t)b * V * d, vis_mid + (size_t)b * V * h, V, d, h, r, g, 0, V, true, smem_raw);
    attn_residual_layernorm_store(Y, erows, tile, gamma, beta, b, q0, Lq, h, eps, out);
}

template <typename In, int T>
static void launch_attn_mfma(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                             const float* beta, int B, int L, int V, int d, int h, float eps, float* out, hipStream_t s) {
    using P = const typename In::T*;
    hipLaunchKernelGGL((attn_fuse_mfma_kernel<In, T>), dim3((L + 15) / 16, B), dim3(64), kStageBytes + sizeof(float) * 16 * (h + 4), s, (P)vis, (P)txt, (P)vis_mid,
                       (P)enc_x, gamma, beta, L, V, d, h, eps, out);
}

// ---- many keys: the key-split form --------------------------------------------------------------------------
// The shipped factor layout has V = 36 + 36^2 + 36 + 1 = 1369 keys per image (config/model/vlgae.yaml:40-42).  One wavefront per
// (sentence, 16 words) walking ALL keys is B x ceil(L / 16) wavefronts -- 192 at B = 64 on a 1024-SIMD chip.  Here the keys are cut
// into NC chunks of CK (a multiple of 64): grid (word tile, sentence, chunk), every wave runs the SAME tile body over its chunk and
// leaves the streaming-softmax state of its chunk -- running max m, sum z, unnormalised Y -- as a record in accumulator layout
// ([CT][64 lanes][4] + m[16] + z[16] floats: one 1-KiB store per channel tile); a combine launch (grid (word tile, sentence))
// merges the NC records in chunk order (bit-reproducible) and applies the residual + LayerNorm.
struct AttnSplit {
    int WT, CK, NC;
    AttnSplit(int B, int L, int V, int key_chunk) {
        WT = (L + 15) / 16;
        const int tiles = (V + 63) / 64;
        int per;   // 64-key steps per chunk
        if (key_chunk > 0) per = (key_chunk + 63) / 64;
        else if (V <= 256) per = tiles;   // tens of regions (object-only layouts): the one-pass kernels
        else {
            // Cheapest split under a two-term model (us): rounds of 1024 resident wavefronts (one per SIMD: the bf16 kernels keep a whole
            // step of loads in flight in registers) x steps per chunk x ~2 us per 64-key step, plus every wavefront's 16-KiB record
            // written and read once (~8 ns each at a few TB/s).  B = 64, L = 40, V = 1369: 5 steps per chunk (5 chunks, 960 waves in
            // one round); B = 256: 22 steps (one pass per word tile, 768 waves).
            const long waves = (long)B * WT;
            double best = 1e30;
            per = tiles;
            for (int cand = 1; cand <= tiles; ++cand) {
                const long nc = (tiles + cand - 1) / cand, w = waves * nc;
                const double cost = (double)((w + 1023) / 1024) * cand * 2.0 + (double)w * 0.008;
                if (cost < best - 1e-9 || (cost < best + 1e-9 && cand > per)) { best = cost; per = cand; }
            }
        }
        if (per > tiles) per = tiles;
        if (per < 1) per = 1;
        CK = 64 * per;
        NC = (tiles + per - 1) / per;
    }
    static __host__ __device__ size_t record_floats(int h) { return (size_t)(h >> 4) * 256 + 32; }
    size_t records(int B) const { return (size_t)B * WT * NC; }
};

template <typename In>
__global__ __launch_bounds__(64, (kIsBF16<In> ? 1 : 2)) void attn_fuse_split_kernel(   // bf16: one wave per SIMD, its loads a step ahead in registers; fp32: two (<= 256 registers)
    // (no __restrict__: loads through noalias read-only arguments are invariant to the compiler and float past the memory barriers
    //  that pin the issue order of the bf16 tile's load stream)
    const typename In::T* vis, const typename In::T* txt, const typename In::T* vis_mid, int Lq, int V, int d, int h, int CK, int NC,
    int pairs, float* rec) {
    // grid (sentence x chunk pairs padded to a multiple of 8, word tile): workgroups are dealt round-robin over the 8 XCDs, so the
    // word tiles of one (sentence, chunk) -- linear ids x, x + gridDim.x, .. -- share an XCD and read the chunk's keys through ONE L2
    // (speed only; word tile fastest puts them on three XCDs and every key row is fetched from memory three times)
    if ((int)blockIdx.x >= pairs) return;
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / NC, c = blockIdx.x - b * NC, q0 = blockIdx.y * 16;
    const int CT = h >> 4;
    const int qw = min(q0 + r, Lq - 1);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // the transposition image (bf16 features)
    f32x4 Y[kAttnMaxCT];
    float m_run, z_run;
    attn_forward_tile<In, 4>(Y, m_run, z_run, txt + ((size_t)b * (Lq + 1) + 1 + qw) * d, vis + (size_t)b * V * d,
                             vis_mid + (size_t)b * V * h, V, d, h, r, g, c * CK, min(V, (c + 1) * CK), false, smem_raw);
    float* my = rec + (((size_t)b * gridDim.y + blockIdx.y) * NC + c) * AttnSplit::record_floats(h);
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct)
        if (ct < CT) *reinterpret_cast<float4*>(my + ((size_t)ct * 64 + lane) * 4) = make_float4(Y[ct][0], Y[ct][1], Y[ct][2], Y[ct][3]);
    if (g == 0) {
        my[(size_t)CT * 256 + r] = m_run;
        my[(size_t)CT * 256 + 16 + r] = z_run;
    }
}

// Merge the NC chunk records of one (sentence, word tile) in chunk order: on return Y = softmax_v(S) . vis_mid (normalised) and
// (m, z) the softmax statistics of word r over all keys.
__device__ __forceinline__ void attn_combine_records(f32x4 (&Y)[kAttnMaxCT], float& m, float& z, const float* __restrict__ rec0, int NC,
                                                     int h) {
    const int lane = threadIdx.x, r = lane & 15;
    const int CT = h >> 4;
    const size_t RS = AttnSplit::record_floats(h);
    m = neg_infinity();
    for (int c = 0; c < NC; ++c) m = fmaxf(m, rec0[c * RS + (size_t)CT * 256 + r]);
    z = 0.f;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < NC; ++c) {
        const float* rc = rec0 + c * RS;
        const float f = __expf(rc[(size_t)CT * 256 + r] - m);   // every chunk holds at least one key: its max is finite
        z = fmaf(f, rc[(size_t)CT * 256 + 16 + r], z);
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) {
            const float4 y = *reinterpret_cast<const float4*>(rc + ((size_t)min(ct, CT - 1) * 64 + lane) * 4);
            Y[ct][0] = fmaf(f, y.x, Y[ct][0]); Y[ct][1] = fmaf(f, y.y, Y[ct][1]);
            Y[ct][2] = fmaf(f, y.z, Y[ct][2]); Y[ct][3] = fmaf(f, y.w, Y[ct][3]);
        }
    }
    const float zinv = 1.f / z;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) Y[ct] *= zinv;
}

// The NC records of one (sentence, word tile) added up front by MANY wavefronts -- grid (word tile, sentence, group of four channel
// tiles) -- into ONE record of the same layout (unnormalised Y relative to the overall max, its max and sum): the kernels that consume
// it (B x word tiles wavefronts, each with a LayerNorm or its adjoint behind it) then read 16 KiB instead of NC x 16 KiB behind a
// dependent chain.  Chunk order, fused multiply-adds: the same bits wherever the merge runs.
__global__ __launch_bounds__(64) void attn_merge_records_kernel(const float* __restrict__ rec, int NC, int h, float* __restrict__ merged) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int CT = h >> 4, ct0 = 4 * blockIdx.z;
    const size_t RS = AttnSplit::record_floats(h);
    const size_t tile_id = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    const float* rec0 = rec + tile_id * NC * RS;
    float m = neg_infinity();
    for (int c = 0; c < NC; ++c) m = fmaxf(m, rec0[c * RS + (size_t)CT * 256 + r]);
    float z = 0.f;
    float4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int c = 0; c < NC; ++c) {
        const float* rc = rec0 + c * RS;
        const float f = __expf(rc[(size_t)CT * 256 + r] - m);
        z = fmaf(f, rc[(size_t)CT * 256 + 16 + r], z);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float4 y = *reinterpret_cast<const float4*>(rc + ((size_t)min(ct0 + i, CT - 1) * 64 + lane) * 4);
            acc[i].x = fmaf(f, y.x, acc[i].x); acc[i].y = fmaf(f, y.y, acc[i].y);
            acc[i].z = fmaf(f, y.z, acc[i].z); acc[i].w = fmaf(f, y.w, acc[i].w);
        }
    }
    float* out = merged + tile_id * RS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (ct0 + i < CT) *reinterpret_cast<float4*>(out + ((size_t)(ct0 + i) * 64 + lane) * 4) = acc[i];
    if (blockIdx.z == 0 && g == 0) {
        out[(size_t)CT * 256 + r] = m;
        out[(size_t)CT * 256 + 16 + r] = z;
    }
}

template <typename In>
__global__ __launch_bounds__(64) void attn_fuse_combine_kernel(const float* __restrict__ rec, int NC, const typename In::T* __restrict__ enc_x,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, int Lq, int h,
                                                               float eps, float* __restrict__ out) {
    const int lane = threadIdx.x;
    const int b = blockIdx.y, q0 = blockIdx.x * 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);
    float4 erows[16];
    attn_load_residual_rows<In>(erows, enc_x, b, q0, Lq, h, lane);
    f32x4 Y[kAttnMaxCT];
    float m, z;
    attn_combine_records(Y, m, z, rec + ((size_t)b * gridDim.x + blockIdx.x) * NC * AttnSplit::record_floats(h), NC, h);
    attn_residual_layernorm_store(Y, erows, tile, gamma, beta, b, q0, Lq, h, eps, out);
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

// Gradient element types: float32, or bfloat16 (rounded to nearest even once, from the fp32 accumulators -- the same bits as an fp32
// result followed by a cast, without the fp32 round trip through HBM and the cast launch).
struct F32Grad { using T = float; };
struct BF16Grad { using T = uint16_t; };
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(uint16_t* p, float4 v) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    const bf16x2 lo = {(__bf16)v.x, (__bf16)v.y}, hi = {(__bf16)v.z, (__bf16)v.w};   // v_cvt_pk_bf16_f32
    uint2 u;
    u.x = __builtin_bit_cast(uint32_t, lo);
    u.y = __builtin_bit_cast(uint32_t, hi);
    *reinterpret_cast<uint2*>(p) = u;
}

// txt as the regions kernel's A fragments of its second product, bf16 [B][word tile][feature tile][64 lanes][4] (lane (r', g')
// element e <-> feature 16 ft + r', word 16 wt + 4 g' + e): one 16-word tile per call.
__device__ __forceinline__ void attn_write_txtT(const uint16_t* __restrict__ txt, int b, int q0, int Lq, int d, int Lp,
                                                uint16_t* __restrict__ txtT) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int FT = d >> 4;
    const uint16_t* row = txt + ((size_t)b * (Lq + 1) + 1 + min(q0 + r, Lq - 1)) * d + 4 * g;
    for (int ft = 0; ft < FT; ++ft) {
        const uint2 u = *reinterpret_cast<const uint2*>(row + 16 * ft);
        uint16_t* o = txtT + ((((size_t)b * (Lp >> 4) + (q0 >> 4)) * FT + ft) * 64 + (r >> 2) * 16 + 4 * g) * 4 + (r & 3);
        o[0] = (uint16_t)(u.x & 0xffffu);
        o[4] = (uint16_t)(u.x >> 16);
        o[8] = (uint16_t)(u.y & 0xffffu);
        o[12] = (uint16_t)(u.y >> 16);
    }
}

// ---- LayerNorm adjoint of one 16-word tile, in accumulator layout (lane (r,g), register (ct,n) <-> word r, channel 16ct+4g+n) ----
// In: Y = M = softmax . vis_mid (normalised).  Out: DY = dy (zero in tiles past h / 16), Dsum = sum_c dy * M; writes the dy rows
// (fp32, [B,L,h]: d_enc_x itself when the gradients are fp32, a scratch beside a bf16 d_enc_x otherwise) and this tile's partial
// rows [d_gamma | d_beta].  tile_x / tile_g: [16][h + 4] floats of LDS each.
// enc_x / dout rows are read here, not at kernel start: 128 more live registers across the forward tile spill.
template <typename In, typename GOut>
__device__ __forceinline__ void attn_ln_adjoint(f32x4 (&Y)[kAttnMaxCT], f32x4 (&DY)[kAttnMaxCT], float& Dsum,
                                                const typename In::T* __restrict__ enc_x, const float* __restrict__ gamma,
                                                const float* __restrict__ dout, size_t ld_dout_b, size_t ld_dout_l, int b, int q0, int Lq,
                                                int h, float eps, float* tile_x, float* tile_g, float* __restrict__ part_row,
                                                float* __restrict__ dy_rows, typename GOut::T* __restrict__ d_enc,
                                                uint16_t* __restrict__ dyT, int Lp) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int CT = h >> 4;
    const int hp = h + 4, cl = min(4 * lane, h - 4);
    const bool cin = 4 * lane < h;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
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
    {   // d_beta partial: column sums of dout over the live words of the tile
        float4 acc = zero4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float k = q0 + i < Lq ? 1.f : 0.f;
            const float4 v = *reinterpret_cast<const float4*>(tile_g + i * hp + cl);
            acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
        }
        if (cin) *reinterpret_cast<float4*>(part_row + h + 4 * lane) = acc;
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
    Dsum = rstd * (a1 - c1 * a2 - c2 * a3);
    // bf16 features: the sweep's dP = <dy, mid_v> runs on the matrix cores with dy ROUNDED to bf16, so D is taken over the same rounded
    // cotangent, D = sum_c bf16(dy_c) M_c (M = y - x again from the normalised row): dS = P (dP - D) is then the exact softmax adjoint of
    // ONE cotangent -- its rows sum to zero over the keys and, with a single key, vanish -- instead of the difference of two roundings
    // (tools/stress_attn.py: V = 1 gave |d_vis| 0.04 ... 0.12 where the gradient is zero)
    const float sd = 1.f / rstd;
    float dq = 0.f;
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) {
        const float keep = ct < CT ? 1.f : 0.f;
        if constexpr (kIsBF16<In>) {
            const float4 e = *reinterpret_cast<const float4*>(tile_x + r * hp + 16 * min(ct, CT - 1) + 4 * g);   // (x: read before dy takes its place)
            const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const float dyv = keep * rstd * (DY[ct][n] - c1 - Y[ct][n] * c2);
                dq = fmaf((float)(__bf16)dyv, keep * (fmaf(Y[ct][n], sd, mean) - ev[n]), dq);
            }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) DY[ct][n] = keep * rstd * (DY[ct][n] - c1 - Y[ct][n] * c2);   // dy
        if (ct < CT) {
            *reinterpret_cast<float4*>(tile_x + r * hp + 16 * ct + 4 * g) = make_float4(DY[ct][0], DY[ct][1], DY[ct][2], DY[ct][3]);
            if constexpr (kIsBF16<In>) {   // dy as the regions kernel's A fragments, bf16 [B][word tile][channel tile][64 lanes][4]:
                // lane (r', g') element e <-> channel 16 ct + r', word 16 wt + 4 g' + e  -- one contiguous 512-byte read per fragment
                // (all 16 words of the tile are written: those past Lq repeat the last word and meet zero weights)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    dyT[((((size_t)b * (Lp >> 4) + (q0 >> 4)) * CT + ct) * 64 + (r >> 2) * 16 + 4 * g + n) * 4 + (r & 3)] =
                        __builtin_bit_cast(uint16_t, (__bf16)DY[ct][n]);
            }
        }
    }
    if constexpr (kIsBF16<In>) Dsum = group_sum4(dq);
    {   // d_gamma partial (column sums of dout*yhat) and d_enc_x rows, both as whole rows
        float4 acc = zero4;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool lv = q0 + i < Lq;
            const float k = lv ? 1.f : 0.f;
            const float4 v = *reinterpret_cast<const float4*>(tile_g + i * hp + cl);
            acc.x = fmaf(k, v.x, acc.x); acc.y = fmaf(k, v.y, acc.y); acc.z = fmaf(k, v.z, acc.z); acc.w = fmaf(k, v.w, acc.w);
            const float4 dyr = *reinterpret_cast<const float4*>(tile_x + i * hp + cl);
            if (lv && cin) {
                const size_t o = ((size_t)b * Lq + q0 + i) * h + 4 * lane;
                *reinterpret_cast<float4*>(dy_rows + o) = dyr;
                if constexpr (sizeof(typename GOut::T) == 2) st4(d_enc + o, dyr);
            }
        }
        if (cin) *reinterpret_cast<float4*>(part_row + 4 * lane) = acc;
    }
}

// ---- the sweep over the keys [v_begin, v_end) with the word tile's statistics known: P again, dP, dS, d_txt; P^T and dS^T to scratch ----
// dT accumulates d_txt^T[feature 16ft+4g+n][word r] over this key range; the scratch rows of EVERY key of the 64-key steps the range
// covers are written (zeros past v_end), so the steps of all chunks together fill [0, Vp).
// PF: operand tiles in flight ahead of the MFMAs (4 with one wave per SIMD; 2 where two waves per SIMD cover for each other).
template <typename In, int T, int FTM, int PF>
__device__ __forceinline__ void attn_bwd_sweep(f32x4 (&dT)[FTM], const f32x4 (&DY)[kAttnMaxCT], const typename In::T* trow,
                                               const typename In::T* vis_b, const typename In::T* mid_p, int Vall, int d, int h, int r, int g,
                                               int v_begin, int v_end, float m_run, float zinv, float Dsum, float livef,
                                               float* __restrict__ PT, float* __restrict__ DST, int b, int Vp, int Lp, int q0, char* lds) {
    const int CT = h >> 4, FT = d >> 4;
    const int V = v_end;
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft) dT[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    const __amdgpu_buffer_rsrc_t vis_rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)vis_b, 0, (int)(Vall * d * sizeof(typename In::T)), 0x00020000);
    constexpr int NCS = kAttnMaxCT / 2;   // bf16: channel steps of 32 = two channel tiles
    constexpr int NS = (T + 1) / 2;       // bf16: key steps of 32 = two region tiles
    bf16x8 DYh[kIsBF16<In> ? NCS : 1];
    if constexpr (kIsBF16<In>) {
#pragma unroll
        for (int s2 = 0; s2 < NCS; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) DYh[s2][j] = (__bf16)DY[2 * s2 + (j >> 2)][j & 3];   // element j <-> channel 16 (2s + j/4) + 4g + j%4
    }
    for (int v0 = v_begin; v0 < V; v0 += 16 * T) {
        f32x4 S[T], dP[T];
        attn_scores<In, T>(S, trow, vis_b, V, d, v0, r, g);
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                S[t][n] = v0 + 16 * t + 4 * g + n < V ? __expf(S[t][n] - m_run) * zinv : 0.f;   // P
            dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // GEMM 3: dP^T[region][word] = sum_c mid[region][c] * dy[word][c]; A fragment = contiguous bytes of a mid row.
        // All region tiles advance together over the channel tiles, their fragments PF tiles ahead in a register ring.
        const typename In::T* mrow[T];
#pragma unroll
        for (int t = 0; t < T; ++t) mrow[t] = mid_p + (size_t)min(v0 + 16 * t + r, V - 1) * h + 4 * g;
        if constexpr (kIsBF16<In>) {
            // K step s = channel tiles (2s, 2s+1): two 8-byte reads (channels 16(2s) + 4g .., 16(2s+1) + 4g ..) match DYh's element order
            bf16x8 mf[PF + 1][T];
#pragma unroll
            for (int pfi = 0; pfi < PF; ++pfi)
#pragma unroll
                for (int t = 0; t < T; ++t) mf[pfi][t] = frag_ld8x2(mrow[t] + 16 * min(2 * pfi, CT - 1), mrow[t] + 16 * min(2 * pfi + 1, CT - 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s2 = 0; s2 < NCS; ++s2) {
                if (s2 + PF < NCS) {   // compile-time
#pragma unroll
                    for (int t = 0; t < T; ++t)
                        mf[(s2 + PF) % (PF + 1)][t] =
                            frag_ld8x2(mrow[t] + 16 * min(2 * (s2 + PF), CT - 1), mrow[t] + 16 * min(2 * (s2 + PF) + 1, CT - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < T; ++t)   // dy of tiles past CT is zero
                    dP[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mf[s2 % (PF + 1)][t], DYh[s2], dP[t], 0, 0, 0);
            }
        } else {
            float4 mf[PF + 1][T];
#pragma unroll
            for (int pfi = 0; pfi < PF; ++pfi)
#pragma unroll
                for (int t = 0; t < T; ++t) mf[pfi][t] = ld4(mrow[t] + 16 * min(pfi, CT - 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ct = 0; ct < kAttnMaxCT; ++ct) {
                if (ct + PF < kAttnMaxCT) {   // compile-time
#pragma unroll
                    for (int t = 0; t < T; ++t) mf[(ct + PF) % (PF + 1)][t] = ld4(mrow[t] + 16 * min(ct + PF, CT - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < T; ++t) {   // dy of tiles past CT is zero
                    const float4 a = mf[ct % (PF + 1)][t];
                    dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, DY[ct][0], dP[t], 0, 0, 0);
                    dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, DY[ct][1], dP[t], 0, 0, 0);
                    dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, DY[ct][2], dP[t], 0, 0, 0);
                    dP[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, DY[ct][3], dP[t], 0, 0, 0);
                }
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
                // v < Vp by construction; zero for v >= V.  bf16 features: the scratch holds bf16 (what the regions kernel multiplies)
                if constexpr (kIsBF16<In>) {   // bf16 scratch: [b][word tile][key][16 words] (a key's 16 words are one 32-byte row: what the regions kernel reads)
                    const size_t at = ((((size_t)b * (Lp >> 4) + (q0 >> 4)) * Vp + v) << 4) + r;
                    reinterpret_cast<uint16_t*>(PT)[at] = __builtin_bit_cast(uint16_t, (__bf16)p);
                    reinterpret_cast<uint16_t*>(DST)[at] = __builtin_bit_cast(uint16_t, (__bf16)ds);
                } else {
                    PT[((size_t)b * Vp + v) * Lp + q0 + r] = p;
                    DST[((size_t)b * Vp + v) * Lp + q0 + r] = ds;
                }
                vlane[t][n] = min(v, V - 1) * d + r;
            }
        // GEMM 4: d_txt^T[feature][word] += sum_v vis[v][feature] * dS[word][v]; B operand = dS as it sits
        // operand ring PF feature tiles ahead, as in the forward's second contraction
        if constexpr (kIsBF16<In>) {
            bf16x8 dSh[NS];
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 2 * s2 + (j >> 2);
                    dSh[s2][j] = (__bf16)(t < T ? S[t < T ? t : 0][j & 3] : 0.f);
                }
            attn_keys_product_lds<T, FTM, false>(dT, vis_b, d, v0, V, dSh, dSh, lds);
        } else {
            float vv[PF + 1][T][4];
#pragma unroll
            for (int pfi = 0; pfi < PF; ++pfi)
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int n = 0; n < 4; ++n) vv[pfi][t][n] = buf_ld(In{}, vis_rs, vlane[t][n], 16 * min(pfi, FT - 1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ft = 0; ft < FTM; ++ft) {
                if (ft + PF < FTM) {   // compile-time
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            vv[(ft + PF) % (PF + 1)][t][n] = buf_ld(In{}, vis_rs, vlane[t][n], 16 * min(ft + PF, FT - 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < T; ++t)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        dT[ft] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[ft % (PF + 1)][t][n], S[t][n], dT[ft], 0, 0, 0);
            }
        }
    }
}

// ---- bf16 features, key-split: the sweep as one stream of loads a whole 64-key step ahead (see attn_forward_tile_bf16) ----
//   * the key rows of vis (GEMM 1's A fragments: 4 tiles x 4 x 16 bytes per lane = the whole [64 keys][128 features] tile) are
//     written to the LDS image as soon as the scores are issued -- GEMM 4 reads its vis^T fragments back from there
//     (ds_read_b64_tr_b16), nothing is fetched twice -- and their registers take the next step's rows;
//   * GEMM 3's A fragments are 16-byte reads of vis_mid rows in natural K order (channels 32s + 8g ..), the cotangent dy is read
//     from its fp32 rows in that order (not from accumulator layout), so a fragment is one read, not two;
//   * P^T and dS^T leave as bf16.
constexpr int kSweepPitch = 288;                  // bytes per image row: 128 features + 16 pad (8 rows of a half-wave on 8 bank groups)
constexpr int kSweepBytes = 64 * kSweepPitch;     // [64 keys][128 features] bf16

template <int FTM>
__device__ __forceinline__ void attn_bwd_sweep_stream_bf16(f32x4 (&dT)[FTM], const float* dy_row, const uint16_t* trow, const uint16_t* vis_b,
                                                           const uint16_t* mid_p, int Vall, int d, int h, int r, int g, int v_begin,
                                                           int v_end, float m_run, float zinv, float Dsum, float livef, uint16_t* PT,
                                                           uint16_t* DST, int b, int Vp, int Lp, int q0, char* lds) {
    constexpr int T = 4, KJ = 4, NCS = kAttnMaxCT / 2;
    static_assert(FTM == 8, "the streaming sweep serves d <= 128 (one 128-feature tile per key row)");
    const int lane = threadIdx.x, q = (lane >> 2) & 3, p4 = lane & 3;
    const int V = v_end;
    const __amdgpu_buffer_rsrc_t vis_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vis_b, 0, Vall * d * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t mid_rs = __builtin_amdgcn_make_buffer_rsrc((void*)mid_p, 0, Vall * h * 2, 0x00020000);
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft) dT[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 wf0[KJ];   // this lane's word (zeros past a 16-feature tail)
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        const int k = 32 * j + 8 * g;
        const bf16x8 x = frag_ld16(trow + (k < d ? k : 0));
        wf0[j] = k < d ? x : frag_zero();
    }
    bf16x8 DYh[NCS];   // dy of this lane's word, channels 32s + 8g .. + 7 (zeros past h)
#pragma unroll
    for (int s2 = 0; s2 < NCS; ++s2) {
        const int c0 = 32 * s2 + 8 * g;
        const bool in = c0 < h;
        const float4 lo = *reinterpret_cast<const float4*>(dy_row + (in ? c0 : 0)), hi = *reinterpret_cast<const float4*>(dy_row + (in ? c0 + 4 : 0));
        const float v8[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) DYh[s2][j] = (__bf16)(in ? v8[j] : 0.f);
    }
    u32x4 vf[T][KJ];     // next step: key rows of vis
    u32x4 mf[NCS][T];    // next step: key rows of vis_mid, 32 channels per step s
    const int vlane = (r * d + 8 * g) * 2, mlane = (r * h + 8 * g) * 2;
    auto fetch_vis = [&](int v0) {
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < KJ; ++j) vf[t][j] = buf_ld128(vis_rs, vlane + 64 * j, (v0 + 16 * t) * d * 2);
    };
    fetch_vis(v_begin);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s2 = 0; s2 < NCS; ++s2) {
#pragma unroll
        for (int t = 0; t < T; ++t) mf[s2][t] = buf_ld128(mid_rs, mlane + 64 * s2, (v_begin + 16 * t) * h * 2);
        asm volatile("" ::: "memory");
    }
    char* pimg = lds + kSweepBytes;                                // [64 keys][16 words] bf16: P^T, then dS^T
    char* wr = lds + r * kSweepPitch + 16 * g;                     // image[key 16t + r][feature 32j + 8g ..]
    const char* rd = lds + (4 * g + q) * kSweepPitch + p4 * 8;     // transposed reads: rows 4g + q (and + 16) of a 32-key half

    for (int v0 = v_begin; v0 < V; v0 += 16 * T) {
        // ---- GEMM 1 + the image ----
        f32x4 S[T], dP[T];
#pragma unroll
        for (int t = 0; t < T; ++t) S[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KJ; ++j)   // (K step outermost: the T accumulation chains interleave instead of four dependent MFMAs in a row)
#pragma unroll
            for (int t = 0; t < T; ++t)
                S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, vf[t][j]), wf0[j], S[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int j = 0; j < KJ; ++j) *reinterpret_cast<u32x4*>(wr + 16 * t * kSweepPitch + 64 * j) = vf[t][j];
        asm volatile("" ::: "memory");
        fetch_vis(v0 + 16 * T);   // the next step's key rows (zeros past the sentence's end)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int t = 0; t < T; ++t) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                S[t][n] = v0 + 16 * t + 4 * g + n < V ? __expf(S[t][n] - m_run) * zinv : 0.f;   // P
            dP[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- GEMM 3: dP^T[key][word] = sum_c vis_mid[key][c] * dy[word][c] ----
#pragma unroll
        for (int s2 = 0; s2 < NCS; ++s2) {
#pragma unroll
            for (int t = 0; t < T; ++t)
                dP[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, mf[s2][t]), DYh[s2], dP[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < T; ++t) mf[s2][t] = buf_ld128(mid_rs, mlane + 64 * s2, (v0 + 16 * T + 16 * t) * h * 2);
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int v = v0 + 16 * t + 4 * g + n;
                const float p = S[t][n] * livef;                       // words past Lq contribute nothing downstream
                const float ds = p * (dP[t][n] - Dsum);
                S[t][n] = ds;
                // P^T / dS^T of the step as bf16 [64 keys][16 words] images in LDS (zero for keys >= V): they leave below as four contiguous
                // 1-KiB stores (the 32 two-byte stores per step they replace held the vector-memory queue: 60 % of the wave cycles were issue stalls)
                *reinterpret_cast<uint16_t*>(pimg + (16 * t + 4 * g + n) * 32 + r * 2) = __builtin_bit_cast(uint16_t, (__bf16)p);
                *reinterpret_cast<uint16_t*>(pimg + 2048 + (16 * t + 4 * g + n) * 32 + r * 2) = __builtin_bit_cast(uint16_t, (__bf16)ds);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {
            const size_t row0 = ((((size_t)b * (Lp >> 4) + (q0 >> 4)) * Vp + v0) << 4);   // scratch layout [b][word tile][key][16 words]
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const u32x4 xp = *reinterpret_cast<const u32x4*>(pimg + hh * 1024 + lane * 16);
                const u32x4 xs = *reinterpret_cast<const u32x4*>(pimg + 2048 + hh * 1024 + lane * 16);
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(PT + row0) + hh * 1024 + lane * 16) = xp;
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(DST + row0) + hh * 1024 + lane * 16) = xs;
            }
        }
        asm volatile("" ::: "memory");
        bf16x8 dSh[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) dSh[s2][j] = (__bf16)S[2 * s2 + (j >> 2)][j & 3];
        // ---- GEMM 4: d_txt^T[feature][word] += vis^T[feature][key] . dS[key][word], vis^T from the image ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int fg = 0; fg < FTM / 4; ++fg) {
                bf16x8 a[4];
#pragma unroll
                for (int f4 = 0; f4 < 4; ++f4) {
                    const char* base = rd + 32 * hf * kSweepPitch + (4 * fg + f4) * 32;
                    a[f4] = tr_frag(tr_read(base), tr_read(base + 16 * kSweepPitch));
                }
#pragma unroll
                for (int f4 = 0; f4 < 4; ++f4)
                    dT[4 * fg + f4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[f4], dSh[hf], dT[4 * fg + f4], 0, 0, 0);
            }
        asm volatile("" ::: "memory");   // (the next step's image writes stay behind these reads)
    }
}

// One pass (V <= 256): forward tile, LayerNorm adjoint and the sweep over all keys in one wave.
template <typename In, typename GOut, int T, int FTM>
__global__ __launch_bounds__(64) void attn_fuse_bwd_words_kernel(
    const typename In::T* __restrict__ vis, const typename In::T* __restrict__ txt,
    const typename In::T* __restrict__ vis_mid, const typename In::T* __restrict__ enc_x,
    const float* __restrict__ gamma, const float* __restrict__ dout, size_t ld_dout_b, size_t ld_dout_l, int Lq, int V, int d, int h, float eps,
    float* __restrict__ PT, float* __restrict__ DST, int Vp, int Lp, float* __restrict__ part,
    typename GOut::T* __restrict__ d_txt, float* __restrict__ dy_rows, typename GOut::T* __restrict__ d_enc,
    uint16_t* __restrict__ dyT, uint16_t* __restrict__ txtT) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 16;
    const int FT = d >> 4;
    if constexpr (kIsBF16<In>) attn_write_txtT(txt, b, q0, Lq, d, Lp, txtT);
    const int qw = min(q0 + r, Lq - 1);
    const bool live = q0 + r < Lq;
    const int hp = h + 4;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile_x = reinterpret_cast<float*>(smem_raw + kStageBytes);   // [16][hp]: enc_x rows, later dy rows (behind the transposition image)
    float* tile_g = tile_x + 16 * hp;                                    // [16][hp]: dout rows, later dout * yhat rows
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

    if (blockIdx.x == 0)   // the root slot never enters the scores: its gradient is zero
        for (int k = 4 * lane; k < d; k += 256) st4(d_txt + (size_t)b * (Lq + 1) * d + k, zero4);

    const typename In::T* trow = txt + ((size_t)b * (Lq + 1) + 1 + qw) * d;
    const typename In::T* vis_b = vis + (size_t)b * V * d;
    const typename In::T* mid_p = vis_mid + (size_t)b * V * h;
    f32x4 Y[kAttnMaxCT];
    float m_run, z_run;
    attn_forward_tile<In, T>(Y, m_run, z_run, trow, vis_b, mid_p, V, d, h, r, g, 0, V, true, smem_raw);   // Y = M
    const float zinv = 1.f / z_run;

    f32x4 DY[kAttnMaxCT];
    float Dsum;
    attn_ln_adjoint<In, GOut>(Y, DY, Dsum, enc_x, gamma, dout, ld_dout_b, ld_dout_l, b, q0, Lq, h, eps, tile_x, tile_g,
                              part + ((size_t)b * gridDim.x + blockIdx.x) * 2 * h, dy_rows, d_enc, dyT, Lp);

    f32x4 dT[FTM];
    attn_bwd_sweep<In, T, FTM, kAttnPF>(dT, DY, trow, vis_b, mid_p, V, d, h, r, g, 0, V, m_run, zinv, Dsum, live ? 1.f : 0.f, PT, DST, b, Vp, Lp, q0,
                                        smem_raw);
    if (live) {
        typename GOut::T* orow = d_txt + ((size_t)b * (Lq + 1) + 1 + q0 + r) * d + 4 * g;
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft)
            if (ft < FT) st4(orow + 16 * ft, make_float4(dT[ft][0], dT[ft][1], dT[ft][2], dT[ft][3]));
    }
}

// ---- key-split adjoint (many keys): four launches in place of the words kernel --------------------------------
//   1. attn_fuse_split_kernel        : the forward's chunk records again (grid (word tile, sentence, chunk))
//   2. attn_merge_records_kernel, attn_bwd_combine_kernel : merge them -> M, (m, 1/z); LayerNorm adjoint -> dy rows, D, the affine partial rows;
//                                      per-word statistics [m | 1/z | D] to a 48-float record per (sentence, word tile)
//   3. attn_bwd_sweep_kernel         : grid (word tile, sentence, chunk): dy back into accumulator layout, the sweep over the chunk's
//                                      keys; its d_txt^T partial as a record [FT][64 lanes][4]
//   4. attn_bwd_dtxt_kernel          : d_txt rows = the chunk records added in chunk order (bit-reproducible)
template <typename In, typename GOut>
__global__ __launch_bounds__(64) void attn_bwd_combine_kernel(
    const float* __restrict__ rec, int NC, const typename In::T* __restrict__ enc_x, const float* __restrict__ gamma,
    const float* __restrict__ dout, size_t ld_dout_b, size_t ld_dout_l, int Lq, int h, float eps, float* __restrict__ stats,
    float* __restrict__ part, float* __restrict__ dy_rows, typename GOut::T* __restrict__ d_enc,
    const typename In::T* __restrict__ txt, int d, int Lp, uint16_t* __restrict__ dyT, uint16_t* __restrict__ txtT) {
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 16;
    const int hp = h + 4;
    if constexpr (kIsBF16<In>) attn_write_txtT(txt, b, q0, Lq, d, Lp, txtT);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile_x = reinterpret_cast<float*>(smem_raw);
    float* tile_g = tile_x + 16 * hp;
    const size_t tile_id = (size_t)b * gridDim.x + blockIdx.x;
    f32x4 Y[kAttnMaxCT], DY[kAttnMaxCT];
    float m, z, Dsum;
    attn_combine_records(Y, m, z, rec + tile_id * NC * AttnSplit::record_floats(h), NC, h);
    attn_ln_adjoint<In, GOut>(Y, DY, Dsum, enc_x, gamma, dout, ld_dout_b, ld_dout_l, b, q0, Lq, h, eps, tile_x, tile_g,
                              part + tile_id * 2 * h, dy_rows, d_enc, dyT, Lp);
    if (g == 0) {
        float* st = stats + tile_id * 48;
        st[r] = m;
        st[16 + r] = 1.f / z;
        st[32 + r] = Dsum;
    }
}

template <typename In, int FTM>
__global__ __launch_bounds__(64, (FTM <= 8 && !kIsBF16<In> ? 2 : 1)) void attn_bwd_sweep_kernel(   // fp32, d <= 128: two waves per SIMD; bf16: one, its loads a step ahead
    // (no __restrict__ on the inputs: see attn_fuse_split_kernel)
    const typename In::T* vis, const typename In::T* txt, const typename In::T* vis_mid, const float* dy_rows, const float* stats, int Lq, int V,
    int d, int h, int CK, int NC, int pairs, float* PT, float* DST, int Vp, int Lp, float* dtrec) {
    if ((int)blockIdx.x >= pairs) return;   // grid as attn_fuse_split_kernel's
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / NC, c = blockIdx.x - b * NC, q0 = blockIdx.y * 16;
    const int CT = h >> 4, FT = d >> 4;
    const int qw = min(q0 + r, Lq - 1);
    const size_t tile_id = (size_t)b * gridDim.y + blockIdx.y;
    const float* st = stats + tile_id * 48;
    const float m_run = st[r], zinv = st[16 + r], Dsum = st[32 + r];
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // the transposition image (bf16 features)
    f32x4 dT[FTM];
    if constexpr (kIsBF16<In> && FTM == 8) {
        // rows past Lq repeat the last word's (as in the one-pass kernel); their P is zeroed by livef
        attn_bwd_sweep_stream_bf16<FTM>(dT, dy_rows + ((size_t)b * Lq + qw) * h, txt + ((size_t)b * (Lq + 1) + 1 + qw) * d,
                                        vis + (size_t)b * V * d, vis_mid + (size_t)b * V * h, V, d, h, r, g, c * CK, min(V, (c + 1) * CK), m_run,
                                        zinv, Dsum, q0 + r < Lq ? 1.f : 0.f, reinterpret_cast<uint16_t*>(PT), reinterpret_cast<uint16_t*>(DST), b,
                                        Vp, Lp, q0, smem_raw);
    } else {
        f32x4 DY[kAttnMaxCT];
        const float* dyr = dy_rows + ((size_t)b * Lq + qw) * h + 4 * g;
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) {
            const float4 v = *reinterpret_cast<const float4*>(dyr + 16 * min(ct, CT - 1));
            const float keep = ct < CT ? 1.f : 0.f;
            DY[ct] = f32x4{keep * v.x, keep * v.y, keep * v.z, keep * v.w};
        }
        attn_bwd_sweep<In, 4, FTM, 2>(dT, DY, txt + ((size_t)b * (Lq + 1) + 1 + qw) * d, vis + (size_t)b * V * d,
                                   vis_mid + (size_t)b * V * h, V, d, h, r, g, c * CK, min(V, (c + 1) * CK), m_run, zinv, Dsum,
                                   q0 + r < Lq ? 1.f : 0.f, PT, DST, b, Vp, Lp, q0, smem_raw);
    }
    float* my = dtrec + (tile_id * NC + c) * (size_t)FT * 256;
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft)
        if (ft < FT) *reinterpret_cast<float4*>(my + ((size_t)ft * 64 + lane) * 4) = make_float4(dT[ft][0], dT[ft][1], dT[ft][2], dT[ft][3]);
}

template <typename GOut>
__global__ __launch_bounds__(64) void attn_bwd_dtxt_kernel(const float* __restrict__ dtrec, int NC, int Lq, int d,
                                                           typename GOut::T* __restrict__ d_txt) {   // grid (word tile, sentence, feature tile)
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * 16, ft = blockIdx.z;
    const int FT = d >> 4;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0 && ft == 0)   // the root slot never enters the scores: its gradient is zero
        for (int k = 4 * lane; k < d; k += 256) st4(d_txt + (size_t)b * (Lq + 1) * d + k, zero4);
    const float* rec0 = dtrec + ((size_t)b * gridDim.x + blockIdx.x) * NC * (size_t)FT * 256 + ((size_t)ft * 64 + lane) * 4;
    float4 acc = zero4;
#pragma unroll 4
    for (int c = 0; c < NC; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(rec0 + (size_t)c * FT * 256);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (q0 + r < Lq) st4(d_txt + ((size_t)b * (Lq + 1) + 1 + q0 + r) * d + 4 * g + 16 * ft, acc);
}

// bf16 features: the same two products on v_mfma_f32_16x16x16_bf16 (lane (r, g) holds k = 4g .. 4g+3 = the chunk's words 4g + e --
// the fp32 form's four single-k MFMAs as one), A fragments = 8 contiguous bytes of the dy^T / txt^T scratch rows (the one-element
// reads of the fp32 form are one vector-memory instruction per value: 96 per 16 words against 24 here).  The cotangent operands
// (dy, P, dS) enter as single bf16 values.
template <typename GOut, int FTM>
__global__ __launch_bounds__(64) void attn_bwd_regions_bf16_kernel(
    const uint16_t* __restrict__ txtF, const uint16_t* __restrict__ dyF, const float* __restrict__ PT, const float* __restrict__ DST,
    int Vp, int Lp, int V, int d, int h, typename GOut::T* __restrict__ d_mid, typename GOut::T* __restrict__ d_vis) {
    using GT = typename GOut::T;
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, v0 = blockIdx.x * 16, v = v0 + r;   // v < Vp
    const int CT = h >> 4, FT = d >> 4, WT = Lp >> 4;
    f32x4 dM[kAttnMaxCT], dV[FTM];
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct) dM[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft) dV[ft] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bf16 scratch [b][word tile][key][16 words]: a wavefront's 16 keys x 32 bytes of one word tile are 512 contiguous bytes
    const uint16_t* prow = reinterpret_cast<const uint16_t*>(PT) + (((size_t)b * WT * Vp + v) << 4) + 4 * g;
    const uint16_t* srow = reinterpret_cast<const uint16_t*>(DST) + (((size_t)b * WT * Vp + v) << 4) + 4 * g;
    const uint16_t* afrag = dyF + (size_t)b * WT * CT * 256 + lane * 4;   // fragment (word tile, channel tile): 64 lanes x 8 contiguous bytes
    const uint16_t* tfrag = txtF + (size_t)b * WT * FT * 256 + lane * 4;
    for (int wt = 0; wt < WT; ++wt) {
        const v4i16 pb = *reinterpret_cast<const v4i16*>(prow + (((size_t)wt * Vp) << 4));
        const v4i16 sb = *reinterpret_cast<const v4i16*>(srow + (((size_t)wt * Vp) << 4));
        v4i16 af[kAttnMaxCT], tf[FTM];
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) af[ct] = *reinterpret_cast<const v4i16*>(afrag + ((size_t)wt * CT + min(ct, CT - 1)) * 256);
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft) tf[ft] = *reinterpret_cast<const v4i16*>(tfrag + ((size_t)wt * FT + min(ft, FT - 1)) * 256);
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct) dM[ct] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af[ct], pb, dM[ct], 0, 0, 0);
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft) dV[ft] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(tf[ft], sb, dV[ft], 0, 0, 0);
    }
    // Results leave as whole rows: accumulator layout (lane (r, g), register n <-> region v0 + r, channel 16 ct + 4g + n) hands every
    // store instruction 16 rows x 8..16 bytes; through a wave-private LDS tile [16 regions][width + 8] each instruction writes
    // 1 KiB of contiguous rows instead.
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    GT* tile = reinterpret_cast<GT*>(smem_raw);
    auto write_rows = [&](int width, GT* dst) {   // tile -> dst: row 0 of this wave's 16 regions, rows `width` apart
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const int pitch = width + 8;
        const int lpr = width * (int)sizeof(GT) / 16;   // lanes (16-byte pieces) per row
        for (int idx = lane; idx < 16 * lpr; idx += 64) {
            const int row = idx / lpr, c = idx - row * lpr;
            const uint4 x = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(tile + row * pitch) + c * 16);
            if (v0 + row < V) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(dst + (size_t)row * width) + c * 16) = x;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile is rewritten next)
    };
#pragma unroll
    for (int ct = 0; ct < kAttnMaxCT; ++ct)
        if (ct < CT) st4(tile + r * (h + 8) + 16 * ct + 4 * g, make_float4(dM[ct][0], dM[ct][1], dM[ct][2], dM[ct][3]));
    write_rows(h, d_mid + ((size_t)b * V + v0) * h);
#pragma unroll
    for (int ft = 0; ft < FTM; ++ft)
        if (ft < FT) st4(tile + r * (d + 8) + 16 * ft + 4 * g, make_float4(dV[ft][0], dV[ft][1], dV[ft][2], dV[ft][3]));
    write_rows(d, d_vis + ((size_t)b * V + v0) * d);
}

// wave = sentence x 16 regions.  d_mid^T[channel][region] = sum_w dy[w][channel] * P[w][region]   (A = dy^T, B = P)
//                                d_vis^T[feature][region] = sum_w t[w][feature]  * dS[w][region]  (A = t^T,  B = dS)
// B fragments are 16 contiguous bytes of a P^T / dS^T scratch row (words are the K index); A values are single
// elements of d_enc_x / txt rows, read through buffer descriptors.
template <typename In, typename GOut, int FTM>
__global__ __launch_bounds__(64) void attn_fuse_bwd_regions_kernel(
    const typename In::T* __restrict__ txt, const float* __restrict__ d_enc, const float* __restrict__ PT,
    const float* __restrict__ DST, int Vp, int Lp, int Lq, int V, int d, int h, typename GOut::T* __restrict__ d_mid,
    typename GOut::T* __restrict__ d_vis) {
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
        {
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
    }
    if (v < V) {   // accumulator layout: lane (r,g), register n <-> region v (col r), channel / feature 16ct + 4g + n
        typename GOut::T* mrow = d_mid + ((size_t)b * V + v) * h + 4 * g;
        typename GOut::T* vrow = d_vis + ((size_t)b * V + v) * d + 4 * g;
#pragma unroll
        for (int ct = 0; ct < kAttnMaxCT; ++ct)
            if (ct < CT) st4(mrow + 16 * ct, make_float4(dM[ct][0], dM[ct][1], dM[ct][2], dM[ct][3]));
#pragma unroll
        for (int ft = 0; ft < FTM; ++ft)
            if (ft < FT) st4(vrow + 16 * ft, make_float4(dV[ft][0], dV[ft][1], dV[ft][2], dV[ft][3]));
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
    AttnSplit sp;
    size_t pt_floats, part_floats, rec_floats, stats_floats, dy_floats, tr_floats, bytes;
    AttnBwdPlan(int B, int L, int V, int d, int h, int grad_dtype, int key_chunk) : sp(B, L, V, key_chunk) {
        WT = (L + 15) / 16;
        T = V > 48 ? 4 : (V + 15) / 16;
        Vp = ((V + 16 * T - 1) / (16 * T)) * 16 * T;
        Lp = 16 * WT;
        pt_floats = (size_t)B * Vp * Lp;
        part_floats = (size_t)B * WT * 2 * h;
        // key-split: the forward's chunk records, then (they are dead by then) the d_txt chunk records in the same place
        const size_t per = AttnSplit::record_floats(h) > (size_t)(d >> 4) * 256 ? AttnSplit::record_floats(h) : (size_t)(d >> 4) * 256;
        rec_floats = sp.NC > 1 ? sp.records(B) * per + (size_t)B * WT * AttnSplit::record_floats(h) : 0;   // + the merged record of every tile
        stats_floats = sp.NC > 1 ? (size_t)B * WT * 48 : 0;
        dy_floats = grad_dtype == VLG_BF16 ? (((size_t)B * L * h + 63) & ~(size_t)63) : 0;   // fp32 dy rows beside a bf16 d_enc_x
        tr_floats = ((size_t)B * (h + d) * Lp + 1) / 2;   // bf16 features: dy^T [B][h][Lp] | txt^T [B][d][Lp] as bf16
        bytes = sizeof(float) * (2 * pt_floats + part_floats + rec_floats + stats_floats + dy_floats + tr_floats);
    }
};

}  // namespace vlg

extern "C" {

size_t vlg_attn_fuse_workspace(int B, int L, int V, int h, int key_chunk) {
    if (B < 1 || L < 1 || V < 1 || h < 16) return 0;
    const vlg::AttnSplit sp(B, L, V, key_chunk);
    return sp.NC > 1 ? sizeof(float) * (sp.records(B) + (size_t)B * sp.WT) * vlg::AttnSplit::record_floats(h) : 0;   // chunk records + one merged record per tile
}

size_t vlg_attn_fuse_saved_bytes(int B, int L, int V, int h, int key_chunk) {
    if (B < 1 || L < 1 || V < 1 || h < 16) return 0;
    const vlg::AttnSplit sp(B, L, V, key_chunk);
    return sp.NC > 1 ? sizeof(float) * (size_t)B * sp.WT * vlg::AttnSplit::record_floats(h) : 0;
}

int vlg_attn_fuse(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                  const float* beta, int B, int L, int V, int d, int h, int in_dtype, float eps, int key_chunk, void* ws,
                  size_t ws_bytes, float* saved, float* out_att, float* out, void* stream) {
    using namespace vlg;
    if (B < 0 || L < 1 || V < 1 || d < 1 || h < 1)
        return set_error(VLG_ERR_SHAPE, "attn_fuse: bad shape B=%d L=%d V=%d d=%d h=%d", B, L, V, d, h);
    if (B == 0) return 0;
    if (!vis || !txt || !vis_mid || !enc_x || !gamma || !beta || !out) return set_error(VLG_ERR_ARG, "attn_fuse: null buffer");
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "attn_fuse: B=%d exceeds grid.y", B);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "attn_fuse: in_dtype %d", in_dtype);
    // ---- matrix-core path: one wave per 16 words (per key chunk, when split), no attention map requested ----
    if (!out_att && d % 16 == 0 && h % 16 == 0 && h <= 16 * kAttnMaxCT && (size_t)V * h * 4 < (1u << 31)) {
        const AttnSplit sp(B, L, V, key_chunk);
        if (sp.NC > 1) {   // many keys: chunk records, then the combine
            const size_t need = sizeof(float) * (sp.records(B) + (size_t)B * sp.WT) * AttnSplit::record_floats(h);
            float* merged = saved ? saved : (float*)ws + sp.records(B) * AttnSplit::record_floats(h);
            if (!ws || ws_bytes < need) return set_error(VLG_ERR_WORKSPACE, "attn_fuse: workspace %zu bytes < %zu", ws_bytes, need);
#define VLG_ATTN_SPLIT(INV)                                                                                                      \
    do {                                                                                                                         \
        hipLaunchKernelGGL((attn_fuse_split_kernel<INV>), dim3((B * sp.NC + 7) / 8 * 8, sp.WT), dim3(64), kStageBytes, s,         \
                           (const INV::T*)vis, (const INV::T*)txt, (const INV::T*)vis_mid, L, V, d, h, sp.CK, sp.NC, B * sp.NC,  \
                           (float*)ws);                                                                                          \
        hipLaunchKernelGGL(attn_merge_records_kernel, dim3(sp.WT, B, ((h >> 4) + 3) / 4), dim3(64), 0, s, (const float*)ws, sp.NC, h, \
                           merged);                                                                                              \
        hipLaunchKernelGGL((attn_fuse_combine_kernel<INV>), dim3(sp.WT, B), dim3(64), sizeof(float) * 16 * (h + 4), s,           \
                           (const float*)merged, 1, (const INV::T*)enc_x, gamma, beta, L, h, eps, out);                          \
    } while (0)
            if (in_dtype == VLG_F32) VLG_ATTN_SPLIT(F32In);
            else VLG_ATTN_SPLIT(BF16In);
#undef VLG_ATTN_SPLIT
            return check_launch("attn_fuse_split_kernel");
        }
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


size_t vlg_attn_fuse_backward_workspace(int B, int L, int V, int d, int h, int grad_dtype, int key_chunk) {
    if (B < 1 || L < 1 || V < 1 || d < 16 || h < 16) return 0;
    return vlg::AttnBwdPlan(B, L, V, d, h, grad_dtype, key_chunk).bytes;
}
}  // extern "C"

namespace vlg {

template <typename In, typename GOut, int FTM>
static void attn_backward_launch(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                                 const float* dout, size_t ld_b, size_t ld_l, int B, int L, int V, int d, int h, float eps,
                                 const AttnBwdPlan& p, const float* saved, float* wsf, void* d_vis, void* d_txt, void* d_vis_mid,
                                 void* d_enc_x, float* d_gamma, float* d_beta, hipStream_t s) {
    using P = const typename In::T*;
    using G = typename GOut::T*;
    float *PT = wsf, *DST = PT + p.pt_floats, *part = DST + p.pt_floats, *rec = part + p.part_floats, *stats = rec + p.rec_floats,
          *dy_scratch = stats + p.stats_floats;
    uint16_t* dyT = reinterpret_cast<uint16_t*>(dy_scratch + p.dy_floats);
    uint16_t* txtT = dyT + (size_t)B * h * p.Lp;
    float* dy_rows = sizeof(typename GOut::T) == 2 ? dy_scratch : (float*)d_enc_x;   // fp32 gradients: d_enc_x IS the dy rows
    const size_t lds = sizeof(float) * 32 * (h + 4), lds_words = kStageBytes + lds;
    if (p.sp.NC > 1) {
        const int pairs = B * p.sp.NC;
        const dim3 tiles(p.WT, B), chunks((pairs + 7) / 8 * 8, p.WT);
        const float* merged = saved;   // the forward's merged records, when the caller kept them (vlg_attn_fuse's `saved`)
        if (!merged) {
            float* mine = rec + p.rec_floats - (size_t)B * p.WT * AttnSplit::record_floats(h);
            hipLaunchKernelGGL((attn_fuse_split_kernel<In>), chunks, dim3(64), kStageBytes, s, (P)vis, (P)txt, (P)vis_mid, L, V, d, h, p.sp.CK,
                               p.sp.NC, pairs, rec);
            hipLaunchKernelGGL(attn_merge_records_kernel, dim3(p.WT, B, ((h >> 4) + 3) / 4), dim3(64), 0, s, (const float*)rec, p.sp.NC, h, mine);
            merged = mine;
        }
        hipLaunchKernelGGL((attn_bwd_combine_kernel<In, GOut>), tiles, dim3(64), lds, s, merged, 1, (P)enc_x, gamma, dout,
                           ld_b, ld_l, L, h, eps, stats, part, dy_rows, (G)d_enc_x, (P)txt, d, p.Lp, dyT, txtT);
        hipLaunchKernelGGL((attn_bwd_sweep_kernel<In, FTM>), chunks, dim3(64), (kSweepBytes + 4096 > kStageBytes ? kSweepBytes + 4096 : kStageBytes), s, (P)vis, (P)txt, (P)vis_mid, (const float*)dy_rows,
                           (const float*)stats, L, V, d, h, p.sp.CK, p.sp.NC, pairs, PT, DST, p.Vp, p.Lp, rec);
        hipLaunchKernelGGL((attn_bwd_dtxt_kernel<GOut>), dim3(p.WT, B, d >> 4), dim3(64), 0, s, (const float*)rec, p.sp.NC, L, d, (G)d_txt);
    } else {
#define VLG_WORDS(TT)                                                                                                              \
    hipLaunchKernelGGL((attn_fuse_bwd_words_kernel<In, GOut, TT, FTM>), dim3(p.WT, B), dim3(64), lds_words, s, (P)vis, (P)txt, (P)vis_mid, \
                       (P)enc_x, gamma, dout, ld_b, ld_l, L, V, d, h, eps, PT, DST, p.Vp, p.Lp, part, (G)d_txt, dy_rows, (G)d_enc_x, dyT, txtT)
        switch (p.T) {
            case 1: VLG_WORDS(1); break;
            case 2: VLG_WORDS(2); break;
            case 3: VLG_WORDS(3); break;
            default: VLG_WORDS(4); break;
        }
#undef VLG_WORDS
    }
    if constexpr (kIsBF16<In>)
        hipLaunchKernelGGL((attn_bwd_regions_bf16_kernel<GOut, FTM>), dim3((V + 15) / 16, B), dim3(64),
                           16 * ((h > d ? h : d) + 8) * sizeof(typename GOut::T), s, (const uint16_t*)txtT,
                           (const uint16_t*)dyT, (const float*)PT, (const float*)DST, p.Vp, p.Lp, V, d, h, (G)d_vis_mid, (G)d_vis);
    else
        hipLaunchKernelGGL((attn_fuse_bwd_regions_kernel<In, GOut, FTM>), dim3((V + 15) / 16, B), dim3(64), 0, s, (P)txt,
                           (const float*)dy_rows, (const float*)PT, (const float*)DST, p.Vp, p.Lp, L, V, d, h, (G)d_vis_mid, (G)d_vis);
    hipLaunchKernelGGL(attn_fuse_bwd_affine_kernel, dim3((2 * h + 63) / 64), dim3(1024), 0, s, (const float*)part, B * p.WT, h, d_gamma,
                       d_beta);
}

}  // namespace vlg

extern "C" {
int vlg_attn_fuse_backward(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                           const float* dout, long long ld_dout_b, long long ld_dout_l, int B, int L, int V, int d, int h, int in_dtype,
                           float eps, int key_chunk, int grad_dtype, const float* saved, void* ws, size_t ws_bytes, void* d_vis,
                           void* d_txt, void* d_vis_mid, void* d_enc_x, float* d_gamma, float* d_beta, void* stream) {
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
    if (grad_dtype != VLG_F32 && !(grad_dtype == VLG_BF16 && in_dtype == VLG_BF16))
        return set_error(VLG_ERR_DTYPE, "attn_fuse_backward: grad_dtype %d with in_dtype %d (fp32, or bf16 gradients of bf16 inputs)", grad_dtype, in_dtype);
    const AttnBwdPlan p(B, L, V, d, h, grad_dtype, key_chunk);
    if (!ws || ws_bytes < p.bytes)
        return set_error(VLG_ERR_WORKSPACE, "attn_fuse_backward: workspace %zu bytes < %zu", ws_bytes, p.bytes);
    float* wsf = (float*)ws;
#define VLG_BW(INV, GO)                                                                                                          \
    do {                                                                                                                         \
        if (d > 128)                                                                                                             \
            attn_backward_launch<INV, GO, 16>(vis, txt, vis_mid, enc_x, gamma, dout, (size_t)ld_dout_b, (size_t)ld_dout_l, B, L, V, d, h, \
                                              eps, p, saved, wsf, d_vis, d_txt, d_vis_mid, d_enc_x, d_gamma, d_beta, s);         \
        else                                                                                                                     \
            attn_backward_launch<INV, GO, 8>(vis, txt, vis_mid, enc_x, gamma, dout, (size_t)ld_dout_b, (size_t)ld_dout_l, B, L, V, d, h,  \
                                             eps, p, saved, wsf, d_vis, d_txt, d_vis_mid, d_enc_x, d_gamma, d_beta, s);          \
    } while (0)
    if (in_dtype == VLG_F32) VLG_BW(F32In, F32Grad);
    else if (grad_dtype == VLG_F32) VLG_BW(BF16In, F32Grad);
    else VLG_BW(BF16In, BF16Grad);
#undef VLG_BW
    return check_launch("attn_fuse_backward");
}
}  // extern "C"
