// vlg_ground.hip -- the grounding loss on the fused alignment maxima (gfx950).
//
//   loss_grounding_factor_ce, src/model/joint.py:439-491, consuming what vlg_align.hip's ARGS variant of the alignment
//   kernel leaves behind (max over regions / over queries of every caption x image pair, and where each maximum sits)
//   instead of the [B,A,Q,V] tensor; and its gradient to both feature tensors.
//
//   1. ground_ce_kernel    : log-softmax cross-entropy along the pair axis, both directions with one kernel
//                            (txt2vis: softmax over images of max_v, joint.py:472-476; vis2txt: softmax over captions
//                            of max_q, :478-483).  Overwrites the maxima with w * (p - delta), the loss's derivative
//                            w.r.t. them up to the global factor, and leaves one partial loss per block.
//   2. ground_sum_kernel   : partial losses -> {txt2vis, vis2txt, total} and the two global factors
//                            num / (loss + 1e-6)  (joint.py:477,484-489; denominators are detached there).
//   3. ground_bwd_kernel   : the gradient reaches the features only through the arg-max positions (max), only where
//                            both masks are on (masked_fill_), as  g * other_side_row  (the contraction).  One pair
//                            is ONE row update, so this is sparse work: 2 * B*A*(Q+V) row-FMAs instead of two more
//                            [B*Q, A*V] x [A*V, d] contractions over a 774 MB gradient tensor.
//                            Block = one caption (gradient of txt) or one image (gradient of vis); every output row is
//                            owned by exactly one wave, which adds its terms in a fixed order -- no atomics, and the
//                            order is the CPU oracle's.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

#include <algorithm>

#include "vlg_common.h"
#include "vlg_dp_core.h"   // F32In / BF16In element loaders
#include "vlg_ground.h"

namespace vlg {

constexpr int kCeThreads = 256, kCeMaxY = 8;   // threads per block; most column shares per row of blocks

// Rows i = 0..n-1, columns j < ncols of  x[i][j] = buf[fixed * fixed_stride + i * row_stride + j]  (fixed = blockIdx.x).
// Per column: log-softmax over i, the diagonal term i == fixed enters the loss with weight w[fixed][j].
// The derivative is gated here, once, by the two masks (masked_fill_ passes no gradient): column j of this block must be
// on in self_mask[fixed][j], and the arg-max position of (row i, column j) must be on in other_mask[i][.].
template <typename W>
__global__ __launch_bounds__(kCeThreads) void ground_ce_kernel(float* __restrict__ buf, size_t fixed_stride,
                                                               size_t row_stride, int n, int ncols,
                                                               const W* __restrict__ w, const uint16_t* __restrict__ arg,
                                                               const uint8_t* __restrict__ self_mask,
                                                               const uint8_t* __restrict__ other_mask, int n_other,
                                                               float* __restrict__ partial) {
    __shared__ float red[kCeThreads];
    __shared__ float colv[kCeThreads];
    const int fixed = blockIdx.x, tid = threadIdx.x;
    float* base = buf + (size_t)fixed * fixed_stride;
    float loss = 0.f;   // thread 0 accumulates the block's loss, columns in ascending order
    // blockIdx.y: this block's share of the columns (whole 256-column strips), so that few long rows still cover the chip
    const int strips = (ncols + kCeThreads - 1) / kCeThreads, spb = (strips + gridDim.y - 1) / gridDim.y;
    const int j_begin = blockIdx.y * spb * kCeThreads, j_end = min(ncols, j_begin + spb * kCeThreads);
    for (int j0 = j_begin; j0 < j_end; j0 += kCeThreads) {
        const int cw = min(kCeThreads, j_end - j0), P = kCeThreads / cw;   // P row slices per column
        const int jl = tid % cw, part = tid / cw, j = j0 + jl;
        const bool on = part < P;
        // column maximum
        float m = __uint_as_float(0xff800000u);
        if (on)
            for (int i = part; i < n; i += P) m = fmaxf(m, base[(size_t)i * row_stride + j]);
        red[tid] = m;
        __syncthreads();
        if (tid < cw) {
            float mm = red[tid];
            for (int p = 1; p < P; ++p) mm = fmaxf(mm, red[p * cw + tid]);
            colv[tid] = mm;
        }
        __syncthreads();
        m = colv[jl];
        __syncthreads();
        // sum of exp
        float z = 0.f;
        if (on)
            for (int i = part; i < n; i += P) z += __expf(base[(size_t)i * row_stride + j] - m);
        red[tid] = z;
        __syncthreads();
        if (tid < cw) {
            float zz = red[tid];
            for (int p = 1; p < P; ++p) zz += red[p * cw + tid];   // fixed order
            colv[tid] = zz;
        }
        __syncthreads();
        z = colv[jl];
        __syncthreads();
        const float wt = on ? (w ? (float)w[(size_t)fixed * ncols + j] : 1.f) : 0.f;
        if (tid < cw) red[tid] = -wt * ((base[(size_t)fixed * row_stride + j] - m) - __logf(z));   // joint.py:476-477
        __syncthreads();
        if (tid == 0)
            for (int c = 0; c < cw; ++c) loss += red[c];
        // derivative w.r.t. the maxima, in place
        const float zinv = 1.f / z;
        const float gate_j = on && (!self_mask || self_mask[(size_t)fixed * ncols + j]) ? wt : 0.f;
        if (on)
            for (int i = part; i < n; i += P) {
                const size_t at = (size_t)fixed * fixed_stride + (size_t)i * row_stride + j;
                const bool open_ = !other_mask || other_mask[(size_t)i * n_other + arg[at]];
                buf[at] = open_ ? gate_j * (__expf(buf[at] - m) * zinv - (i == fixed ? 1.f : 0.f)) : 0.f;
            }
        __syncthreads();
    }
    if (tid == 0) partial[(size_t)fixed * gridDim.y + blockIdx.y] = loss;
}

// The same cross-entropy with the strip staged in LDS: a block owns `cw` columns of one caption / image, brings the
// [n x cw] strip of maxima in with one coalesced pass (every thread has ~24 independent loads in flight, where the kernel
// above walks a column with one dependent stream per thread and is latency-bound at one block per CU), takes the column
// statistics from LDS, and writes the derivative back with one more coalesced pass.  Same summation orders per column.
constexpr int kCeTileThreads = 512;

template <typename W>
__device__ __forceinline__ void ground_ce_tile_body(float* __restrict__ buf, size_t fixed_stride, size_t row_stride, int n, int ncols,
                                                    int cw_max, const W* __restrict__ w, const uint16_t* __restrict__ arg,
                                                    const uint8_t* __restrict__ self_mask, const uint8_t* __restrict__ other_mask,
                                                    int n_other, float* __restrict__ partial, int by, int ny) {
    extern __shared__ float ce_lds[];
    constexpr int NT = kCeTileThreads;
    const int fixed = blockIdx.x, tid = threadIdx.x;
    const int j0 = by * cw_max, cw = min(cw_max, ncols - j0), pitch = cw_max | 1;
    float* tile = ce_lds;                        // [n][pitch]
    float* red = tile + (size_t)n * pitch;       // [NT]
    float* colm = red + NT;                      // [cw_max] column maximum
    float* colz = colm + cw_max;                 // [cw_max] 1 / sum of exp
    float* colg = colz + cw_max;                 // [cw_max] gate * weight
    float* base = buf + (size_t)fixed * fixed_stride + j0;
    const int total = n * cw, di = NT / cw, dj = NT - di * cw;
    {   // strip in: element e = i * cw + jl, e = tid, tid + NT, ...
        int i = tid / cw, jl = tid - i * cw;
        for (int e = tid; e < total; e += NT) {
            tile[i * pitch + jl] = base[(size_t)i * row_stride + jl];
            i += di;
            jl += dj;
            if (jl >= cw) { jl -= cw; ++i; }
        }
    }
    __syncthreads();
    const int P = NT / cw, jl = tid % cw, part = tid / cw;   // P row slices per column
    const bool on = part < P;
    float m = __uint_as_float(0xff800000u);
    if (on)
        for (int i = part; i < n; i += P) m = fmaxf(m, tile[i * pitch + jl]);
    red[tid] = m;
    __syncthreads();
    if (tid < cw) {
        float mm = red[tid];
        for (int p = 1; p < P; ++p) mm = fmaxf(mm, red[p * cw + tid]);
        colm[tid] = mm;
    }
    __syncthreads();
    m = colm[jl];
    float z = 0.f;
    if (on)
        for (int i = part; i < n; i += P) z += __expf(tile[i * pitch + jl] - m);
    red[tid] = z;
    __syncthreads();
    if (tid < cw) {
        float zz = red[tid];
        for (int p = 1; p < P; ++p) zz += red[p * cw + tid];   // fixed order
        const int j = j0 + tid;
        const float wt = w ? (float)w[(size_t)fixed * ncols + j] : 1.f;
        colz[tid] = 1.f / zz;
        colg[tid] = (!self_mask || self_mask[(size_t)fixed * ncols + j]) ? wt : 0.f;
        red[tid] = -wt * ((tile[fixed * pitch + tid] - colm[tid]) - __logf(zz));   // joint.py:476-477
    }
    __syncthreads();
    if (tid == 0) {
        float loss = 0.f;
        for (int c = 0; c < cw; ++c) loss += red[c];   // columns in ascending order
        partial[(size_t)fixed * ny + by] = loss;
    }
    {   // derivative w.r.t. the maxima, in place
        const uint16_t* abase = arg + (size_t)fixed * fixed_stride + j0;
        int i = tid / cw, c = tid - i * cw;
        for (int e = tid; e < total; e += NT) {
            const size_t at = (size_t)i * row_stride + c;
            const bool open_ = !other_mask || other_mask[(size_t)i * n_other + abase[at]];
            base[at] = open_ ? colg[c] * (__expf(tile[i * pitch + c] - colm[c]) * colz[c] - (i == fixed ? 1.f : 0.f)) : 0.f;
            i += di;
            c += dj;
            if (c >= cw) { c -= cw; ++i; }
        }
    }
}

template <typename W>
__global__ __launch_bounds__(kCeTileThreads) void ground_ce_tile_kernel(float* __restrict__ buf, size_t fixed_stride, size_t row_stride,
                                                                        int n, int ncols, int cw_max, const W* __restrict__ w,
                                                                        const uint16_t* __restrict__ arg,
                                                                        const uint8_t* __restrict__ self_mask,
                                                                        const uint8_t* __restrict__ other_mask, int n_other,
                                                                        float* __restrict__ partial) {
    ground_ce_tile_body<W>(buf, fixed_stride, row_stride, n, ncols, cw_max, w, arg, self_mask, other_mask, n_other, partial, (int)blockIdx.y,
                           (int)gridDim.y);
}

// Both cross-entropies of the grounding loss in ONE launch (round 3): they are independent, 26 us each, and their strips co-reside
// on a CU -- blockIdx.y < y1: the txt2vis strips (float weights = the marginals), else the vis2txt strips (byte weights = vis_mask).
struct CeSide {
    float* buf; size_t fixed_stride, row_stride; int ncols, cw_max; const void* w; const uint16_t* arg;
    const uint8_t *self_mask, *other_mask; int n_other; float* partial; int ny;
};
__global__ __launch_bounds__(kCeTileThreads) void ground_ce_tile2_kernel(int n, CeSide a, CeSide b) {
    if ((int)blockIdx.y < a.ny)
        ground_ce_tile_body<float>(a.buf, a.fixed_stride, a.row_stride, n, a.ncols, a.cw_max, (const float*)a.w, a.arg, a.self_mask,
                                   a.other_mask, a.n_other, a.partial, (int)blockIdx.y, a.ny);
    else
        ground_ce_tile_body<uint8_t>(b.buf, b.fixed_stride, b.row_stride, n, b.ncols, b.cw_max, (const uint8_t*)b.w, b.arg, b.self_mask,
                                     b.other_mask, b.n_other, b.partial, (int)blockIdx.y - a.ny, b.ny);
}

// sums = {txt2vis, vis2txt, total};  coef = {c1, c2}: d total / d txt2vis, d total / d vis2txt.
// One wave: lane i adds partials i, i+64, ... in order, then a fixed xor tree (same bits every run).
__global__ __launch_bounds__(64) void ground_sum_kernel(const float* __restrict__ part1, const float* __restrict__ part2, int n1,
                                                        int n2, float num_token, float w_v2t, float* __restrict__ sums,
                                                        float* __restrict__ coef) {
    float t2v = 0.f, v2t = 0.f;
    for (int i = threadIdx.x; i < n1; i += 64) t2v += part1[i];
    for (int i = threadIdx.x; i < n2; i += 64) v2t += part2[i];
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) { t2v += __shfl_xor(t2v, k, 64); v2t += __shfl_xor(v2t, k, 64); }
    if (threadIdx.x == 0) {
        const float c1 = num_token / (t2v + 1e-6f), c2 = w_v2t > 0.f ? w_v2t * num_token / (v2t + 1e-6f) : 0.f;
        sums[0] = t2v;
        sums[1] = v2t;
        sums[2] = t2v * c1 + v2t * c2;
        coef[0] = c1;
        coef[1] = c2;
    }
}

// ---- gradient to one side's features ---------------------------------------------------------------------------
// SIDE_TXT: block = caption b, output rows = its queries q, the other side's rows are regions (a, v).
//   gather part  (max over V): for every image a, row q receives  c1 * gV[b,a,q] * vis[a, argV[b,a,q]]
//   scatter part (max over Q): term (a, v) lands on row argQ[b,a,v] with  c2 * gQ[b,a,v] * vis[a, v]
// SIDE_VIS: block = image a, output rows = its regions v, the other side's rows are queries (b, q).
//   scatter part (max over V): term (b, q) lands on row argV[b,a,q] with  c1 * gV[b,a,q] * txt[b, q]
//   gather part  (max over Q): for every caption b, row v receives  c2 * gQ[b,a,v] * txt[b, argQ[b,a,v]]
// (gV / gQ arrive already gated by the masks.)  One term = one row update of d floats, and the machine is bound by the
// latency of the row reads, so the kernel is organised around keeping many of them in flight:
//   * 16 waves per block; output row r belongs to wave r mod 16, for both parts -- one owner per row, so no two waves
//     ever add into the same row and a row's terms are added in program order (run-to-run reproducible, no global atomics);
//   * a term is served by d/4 lanes (16-byte reads), so one instruction serves 64/(d/4) terms (2 at d = 128);
//     kGbStep such instructions are issued back to back before the first result is used;
//   * gather terms of an owned row accumulate in registers; scatter terms first pass an ownership test 64 at a time
//     (one per lane), the survivors are compacted in order into a small per-wave LDS queue, and full queues are
//     drained into the row accumulators in LDS (plain read-add-write: the wave owns the rows).
constexpr int kGbWaves = 16, kGbStep = 16, kGbQueue = 128, kGbScan = 4, kGbDrain = 8;

template <typename In>
__device__ __forceinline__ float4 row_ld4(const typename In::T* p);
template <>
__device__ __forceinline__ float4 row_ld4<F32In>(const float* p) { return *reinterpret_cast<const float4*>(p); }
template <>
__device__ __forceinline__ float4 row_ld4<BF16In>(const uint16_t* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

// NS > 0: this block has at most 16 NS output rows, so a wave owns at most NS of them and the scatter part keeps them in
// registers too (per-lane slot select); NS == 0: any number of rows, scatter drains into the LDS rows by read-add-write.
// COOP (16-row blocks, i.e. one output row per wave, and many scattered terms per block): the scan of the scattered terms
// is done once per block instead of once per wave -- all 1024 lanes read kCoopTerms terms per trip (coalesced), the ones that
// land in this block's rows are compacted IN TERM ORDER into a block queue in LDS (ballots + a 64-entry prefix over
// (batch, wave)), and the waves then pick their own row's entries from LDS and sum the source rows in registers.
#ifndef VLG_COOP_BATCHES
#define VLG_COOP_BATCHES 4
#endif
#ifndef VLG_COOP_CAP
#define VLG_COOP_CAP 8192
#endif
constexpr int kCoopCap = VLG_COOP_CAP, kCoopBatches = VLG_COOP_BATCHES, kCoopTerms = 64 * kGbWaves * kCoopBatches;

template <typename In, bool SIDE_TXT, int NS, bool COOP = false>
__global__ __launch_bounds__(64 * kGbWaves) void ground_bwd_kernel(
    const typename In::T* __restrict__ txt, const typename In::T* __restrict__ vis, const float* __restrict__ gV,
    const uint16_t* __restrict__ argV, const float* __restrict__ gQ, const uint16_t* __restrict__ argQ,
    const float* __restrict__ coef, int B, int Q, int V, int d, int rows_per_block, float* __restrict__ out) {
    const int A = B;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int self = blockIdx.x;   // caption b (SIDE_TXT) or image a
    const int n_all = SIDE_TXT ? Q : V, n_other = SIDE_TXT ? V : Q;
    // this block's output rows [row0, row0 + n_rows) of `self` (blockIdx.y: more rows than the LDS accumulators hold, e.g.
    // the 1369 factors of the shipped obj + rel + attr + img layout)
    const int row0 = blockIdx.y * rows_per_block, n_rows = min(rows_per_block, n_all - row0);
    const typename In::T* other = SIDE_TXT ? vis : txt;   // [B][n_other][d]
    const float* g_gather = SIDE_TXT ? gV : gQ;
    const uint16_t* a_gather = SIDE_TXT ? argV : argQ;
    const float* g_scatter = SIDE_TXT ? gQ : gV;
    const uint16_t* a_scatter = SIDE_TXT ? argQ : argV;
    const float c_gather = SIDE_TXT ? coef[0] : coef[1], c_scatter = SIDE_TXT ? coef[1] : coef[0];
    auto pair_base = [&](int p, int n) -> size_t { return SIDE_TXT ? ((size_t)self * A + p) * n : ((size_t)p * A + self) * n; };

    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* acc = reinterpret_cast<float*>(smem_raw);                       // [n_rows][d]
    int* q_src = reinterpret_cast<int*>(acc + (size_t)n_rows * d) + wave * kGbQueue;   // per-wave queue: source row ...
    int* q_dst = q_src + kGbWaves * kGbQueue;                              // ... destination row ...
    float* q_g = reinterpret_cast<float*>(q_dst + kGbWaves * kGbQueue);    // ... weight
    for (int i = threadIdx.x; i < n_rows * d; i += 64 * kGbWaves) acc[i] = 0.f;
    __syncthreads();

    const int lpt = d >> 2, tps = 64 / lpt;     // lanes per term, terms per instruction
    const int sub = lane / lpt, fl = (lane - sub * lpt) * 4;

    // 64 terms, one per lane (src row, weight; weight 0 = padding), all for destination `dst` (uniform): registers.
    auto gather64 = [&](int src, float g, float4& sum) {
        for (int s0 = 0; s0 < 64; s0 += tps * kGbStep) {
            float4 x[kGbStep];
            float gg[kGbStep];
#pragma unroll
            for (int k = 0; k < kGbStep; ++k) {
                const int i = min(s0 + k * tps + sub, 63);
                const int si = __shfl(src, i, 64);
                gg[k] = s0 + k * tps + sub < 64 ? __shfl(g, i, 64) : 0.f;
                x[k] = row_ld4<In>(other + (size_t)si * d + fl);
            }
#pragma unroll
            for (int k = 0; k < kGbStep; ++k) {
                sum.x = fmaf(gg[k], x[k].x, sum.x); sum.y = fmaf(gg[k], x[k].y, sum.y);
                sum.z = fmaf(gg[k], x[k].z, sum.z); sum.w = fmaf(gg[k], x[k].w, sum.w);
            }
        }
    };
    // up to 64 queued terms (one per lane; weight 0 = padding) into the row accumulators.
    float4 racc[NS > 0 ? NS : 1];
    auto scatter64 = [&](int src, int dst, float g) {
        for (int s0 = 0; s0 < 64; s0 += tps * kGbDrain) {
            float4 x[kGbDrain];
            float gg[kGbDrain];
            int dd[kGbDrain];
#pragma unroll
            for (int k = 0; k < kGbDrain; ++k) {
                const int i = min(s0 + k * tps + sub, 63);
                const int si = __shfl(src, i, 64);
                dd[k] = __shfl(dst, i, 64);
                gg[k] = s0 + k * tps + sub < 64 ? __shfl(g, i, 64) : 0.f;
                x[k] = row_ld4<In>(other + (size_t)si * d + fl);
            }
            if (NS > 0) {
                // the wave owns rows wave, wave + 16, ...: slot = row / 16.  Lanes serving different terms of one
                // instruction may hold different slots, so the select is per lane.
#pragma unroll
                for (int k = 0; k < kGbDrain; ++k) {
                    const int slot = dd[k] >> 4;
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) {
                        const float gs = slot == sl ? gg[k] : 0.f;
                        racc[sl].x = fmaf(gs, x[k].x, racc[sl].x); racc[sl].y = fmaf(gs, x[k].y, racc[sl].y);
                        racc[sl].z = fmaf(gs, x[k].z, racc[sl].z); racc[sl].w = fmaf(gs, x[k].w, racc[sl].w);
                    }
                }
            } else {
                // plain read-add-write on the owner's rows.  The terms served by one instruction may share a row, so
                // they take turns; LDS float atomics would not need that but measured 10x slower (2.7 ms vs 0.25 ms).
#pragma unroll
                for (int k = 0; k < kGbDrain; ++k)
                    for (int turn = 0; turn < tps; ++turn)
                        if (sub == turn && gg[k] != 0.f) {
                            float4* cell = reinterpret_cast<float4*>(acc + (size_t)dd[k] * d + fl);
                            float4 c = *cell;
                            c.x = fmaf(gg[k], x[k].x, c.x); c.y = fmaf(gg[k], x[k].y, c.y);
                            c.z = fmaf(gg[k], x[k].z, c.z); c.w = fmaf(gg[k], x[k].w, c.w);
                            *cell = c;
                        }
            }
        }
    };

    // ---- max-over-V terms first (gather for txt, scatter for vis), then max-over-Q terms: the oracle's order ----
    for (int phase = 0; phase < 2; ++phase) {
        const bool do_gather = SIDE_TXT ? phase == 0 : phase == 1;
        if (do_gather) {
            if (c_gather == 0.f) continue;
            for (int row = wave; row < n_rows; row += kGbWaves) {
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int p0 = 0; p0 < B; p0 += 64) {
                    const int p = min(p0 + lane, B - 1);
                    const size_t at = pair_base(p, n_all) + row0 + row;
                    const int src = p * n_other + a_gather[at];
                    const float g = p0 + lane < B ? c_gather * g_gather[at] : 0.f;
                    gather64(src, g, sum);
                }
                // the tps sub-terms of an instruction each hold a partial sum of the row: fold them in a fixed order
                for (int k = lpt; k < 64; k <<= 1) {
                    sum.x += __shfl_xor(sum.x, k, 64); sum.y += __shfl_xor(sum.y, k, 64);
                    sum.z += __shfl_xor(sum.z, k, 64); sum.w += __shfl_xor(sum.w, k, 64);
                }
                if (sub == 0) {
                    float4* cell = reinterpret_cast<float4*>(acc + (size_t)row * d + fl);
                    float4 c = *cell;
                    c.x += sum.x; c.y += sum.y; c.z += sum.z; c.w += sum.w;
                    *cell = c;
                }
            }
        } else {
            if (c_scatter == 0.f) continue;
            const int n_terms = B * n_other;
            if (COOP) {
                static_assert(!COOP || (NS == 0 && kGbWaves * kCoopBatches <= 64 && kCoopCap >= 2 * kCoopTerms), "one prefix lane per (batch, wave)");
                int* bq_src = reinterpret_cast<int*>(acc + (size_t)n_rows * d) + 3 * kGbWaves * kGbQueue;   // [kCoopCap] source row (after the per-wave lists)
                int* bq_lr = bq_src + kCoopCap;                                  // [kCoopCap] local destination row
                float* bq_g = reinterpret_cast<float*>(bq_lr + kCoopCap);        // [kCoopCap] weight
                int* cnt = reinterpret_cast<int*>(bq_g + kCoopCap);              // [64] hits per (batch, wave) of the trip
                float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
                int queued = 0, filled = 0;   // this wave's list / the block queue (both uniform)
                auto drain = [&]() {
                    for (int i0 = 0; i0 < filled; i0 += 64) {
                        const int i = i0 + lane;
                        const bool mine = i < filled && bq_lr[min(i, kCoopCap - 1)] == wave;
                        const unsigned long long hits = __ballot(mine);
                        if (hits == 0) continue;
                        if (mine) {
                            const int at = queued + __popcll(hits & ((1ull << lane) - 1ull));
                            q_src[at] = bq_src[i];
                            q_g[at] = bq_g[i];
                        }
                        queued += __popcll(hits);
                        if (queued >= 64) {
                            gather64(q_src[lane], q_g[lane], sum);
                            queued -= 64;
                            if (lane < queued) {
                                const int s_ = q_src[64 + lane];
                                const float g_ = q_g[64 + lane];
                                q_src[lane] = s_; q_g[lane] = g_;
                            }
                        }
                    }
                };
                for (int t0 = 0; t0 < n_terms; t0 += kCoopTerms) {
                    int lr_[kCoopBatches];
                    float gs_[kCoopBatches];
                    unsigned long long hb[kCoopBatches];
#pragma unroll
                    for (int u = 0; u < kCoopBatches; ++u) {
                        const int tt = t0 + u * 64 * kGbWaves + (int)threadIdx.x, t = min(tt, n_terms - 1);
                        const int p = t / n_other, pos = t - p * n_other;
                        const size_t at = pair_base(p, n_other) + pos;
                        lr_[u] = (int)a_scatter[at] - row0;
                        gs_[u] = tt < n_terms ? c_scatter * g_scatter[at] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < kCoopBatches; ++u) {
                        hb[u] = __ballot(lr_[u] >= 0 && lr_[u] < n_rows && gs_[u] != 0.f);
                        if (lane == 0) cnt[u * kGbWaves + wave] = __popcll(hb[u]);
                    }
                    __syncthreads();
                    int incl = lane < kGbWaves * kCoopBatches ? cnt[lane] : 0;   // lane <-> (batch, wave) in term order; inclusive prefix
#pragma unroll
                    for (int k = 1; k < 64; k <<= 1) {
                        const int o = __shfl_up(incl, k, 64);
                        if (lane >= k) incl += o;
                    }
                    const int total = __shfl(incl, kGbWaves * kCoopBatches - 1, 64);
#pragma unroll
                    for (int u = 0; u < kCoopBatches; ++u) {
                        const int e = u * kGbWaves + wave;
                        const int base = filled + __shfl(incl, e, 64) - __popcll(hb[u]);
                        if ((hb[u] >> lane) & 1ull) {
                            const int at = base + __popcll(hb[u] & ((1ull << lane) - 1ull));
                            bq_src[at] = min(t0 + u * 64 * kGbWaves + (int)threadIdx.x, n_terms - 1);
                            bq_lr[at] = lr_[u];
                            bq_g[at] = gs_[u];
                        }
                    }
                    filled += total;
                    __syncthreads();
                    if (filled > kCoopCap - kCoopTerms) {
                        drain();
                        filled = 0;
                        __syncthreads();
                    }
                }
                drain();
                if (queued > 0) gather64(lane < queued ? q_src[lane] : 0, lane < queued ? q_g[lane] : 0.f, sum);
                for (int k = lpt; k < 64; k <<= 1) {
                    sum.x += __shfl_xor(sum.x, k, 64); sum.y += __shfl_xor(sum.y, k, 64);
                    sum.z += __shfl_xor(sum.z, k, 64); sum.w += __shfl_xor(sum.w, k, 64);
                }
                if (sub == 0 && wave < n_rows) {
                    float4* cell = reinterpret_cast<float4*>(acc + (size_t)wave * d + fl);
                    float4 c = *cell;
                    c.x += sum.x; c.y += sum.y; c.z += sum.z; c.w += sum.w;
                    *cell = c;
                }
                continue;
            }
#pragma unroll
            for (int sl = 0; sl < (NS > 0 ? NS : 1); ++sl) racc[sl] = make_float4(0.f, 0.f, 0.f, 0.f);
            int queued = 0;   // uniform
            // kGbScan batches of 64 terms per trip: all their index / weight reads are in flight together (one batch per
            // trip made the scan itself the bottleneck: a memory round trip per 64 terms, 16 waves each scanning all terms)
            for (int t00 = 0; t00 < n_terms; t00 += 64 * kGbScan) {
                int rows_[kGbScan];
                float gs_[kGbScan];
#pragma unroll
                for (int u = 0; u < kGbScan; ++u) {
                    const int t = min(t00 + 64 * u + lane, n_terms - 1);
                    const int p = t / n_other, pos = t - p * n_other;
                    const size_t at = pair_base(p, n_other) + pos;
                    rows_[u] = a_scatter[at];
                    gs_[u] = t00 + 64 * u + lane < n_terms ? c_scatter * g_scatter[at] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < kGbScan; ++u) {
                    const int t = t00 + 64 * u + lane;
                    const int lr = rows_[u] - row0;   // local row; owned if in this block's range and lr mod 16 == wave
                    const int row = (lr >= 0 && lr < n_rows && (lr & (kGbWaves - 1)) == wave && gs_[u] != 0.f) ? lr : -1;
                    const unsigned long long hits = __ballot(row >= 0);
                    if (hits == 0) continue;
                    if (row >= 0) {   // compact in term order behind what is already queued
                        const int at = queued + __popcll(hits & ((1ull << lane) - 1ull));
                        q_src[at] = t;   // flat row index on the other side = p * n_other + pos
                        q_dst[at] = row;
                        q_g[at] = gs_[u];
                    }
                    queued += __popcll(hits);
                    if (queued >= 64) {
                        scatter64(q_src[lane], q_dst[lane], q_g[lane]);
                        queued -= 64;
                        if (lane < queued) {   // move the tail to the front (tail < 64 entries: read before write per lane)
                            const int s_ = q_src[64 + lane], d_ = q_dst[64 + lane];
                            const float g_ = q_g[64 + lane];
                            q_src[lane] = s_; q_dst[lane] = d_; q_g[lane] = g_;
                        }
                    }
                }
            }
            if (queued > 0) scatter64(lane < queued ? q_src[lane] : 0, lane < queued ? q_dst[lane] : 0, lane < queued ? q_g[lane] : 0.f);
            if (NS > 0) {
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    float4 sum = racc[sl];
                    for (int k = lpt; k < 64; k <<= 1) {   // fold the sub-term partials in a fixed order
                        sum.x += __shfl_xor(sum.x, k, 64); sum.y += __shfl_xor(sum.y, k, 64);
                        sum.z += __shfl_xor(sum.z, k, 64); sum.w += __shfl_xor(sum.w, k, 64);
                    }
                    const int row = wave + kGbWaves * sl;
                    if (sub == 0 && row < n_rows) {
                        float4* cell = reinterpret_cast<float4*>(acc + (size_t)row * d + fl);
                        float4 c = *cell;
                        c.x += sum.x; c.y += sum.y; c.z += sum.z; c.w += sum.w;
                        *cell = c;
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x * 4; i < n_rows * d; i += 64 * kGbWaves * 4)
        *reinterpret_cast<float4*>(out + ((size_t)self * n_all + row0) * d + i) = *reinterpret_cast<const float4*>(acc + i);
}

template <typename In>
static int launch_bwd(const void* txt, const void* vis, const float* gV, const uint16_t* argV, const float* gQ,
                      const uint16_t* argQ, const float* coef, int B, int Q, int V, int d, float* g_txt, float* g_vis,
                      hipStream_t s, bool max_q_terms = true) {
    using P = const typename In::T*;
    const size_t queue = (size_t)kGbWaves * kGbQueue * 12;
    for (int side = 0; side < 2; ++side) {
        float* out = side == 0 ? g_txt : g_vis;
        if (!out) continue;
        const int rows = side == 0 ? Q : V;
        // output rows per block: what the LDS accumulators hold (128 KB with the queues), in multiples of 16, evenly split
        const int cap = (int)((128 * 1024 - queue) / (sizeof(float) * d)) & ~15;
        if (cap < 16) return set_error(VLG_ERR_SHAPE, "grounding_loss: d=%d exceeds the LDS accumulator budget", d);
        int ny = (rows + cap - 1) / cap;
        // more row chunks per caption / image: small batches need them to cover the chip, and with enough scattered terms
        // one row per wave plus the block-level scan wins outright (B = 64, V = 1369: caption side 2.0 -> 0.42 ms; config-2:
        // 1.22 -> 1.18 ms for the whole loss; the same chunking without the block-level scan: 1.59 ms)
        // terms scattered onto this side's rows: the max-over-Q ones for captions (absent when their factor is 0), max-over-V for images
        const size_t n_scatter = side == 0 ? (max_q_terms ? (size_t)B * V : 0) : (size_t)B * Q;
        if (n_scatter >= 2 * (size_t)kCoopTerms) ny = (rows + 15) / 16;   // one row per wave + block-level scan (COOP)
        else if (B * ny < 256) ny = std::max(ny, std::min((rows + 15) / 16, (512 + B - 1) / B));
        const int rpb = ny == 1 ? rows : (((rows + ny - 1) / ny) + 15) & ~15;
        // one row per wave and a long list of scattered terms: scan them once per block (COOP)
        const bool coop = (rpb == kGbWaves || rows <= kGbWaves) && n_scatter >= 2 * (size_t)kCoopTerms;
        const size_t lds = sizeof(float) * (size_t)rpb * d + queue + (coop ? (size_t)kCoopCap * 12 + 256 : 0);
        void (*k)(P, P, const float*, const uint16_t*, const float*, const uint16_t*, const float*, int, int, int, int, int, float*);
        // register slots pay off up to 3 rows per wave; with 6 the per-lane selects cost more than the LDS round trips
        if (side == 0) k = coop ? ground_bwd_kernel<In, true, 0, true> : rows <= 48 ? ground_bwd_kernel<In, true, 3> : ground_bwd_kernel<In, true, 0>;
        else k = coop ? ground_bwd_kernel<In, false, 0, true> : rows <= 48 ? ground_bwd_kernel<In, false, 3> : ground_bwd_kernel<In, false, 0>;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(k, dim3(B, (rows + rpb - 1) / rpb), dim3(64 * kGbWaves), lds, s, (P)txt, (P)vis, gV, argV, gQ, argQ, coef,
                           B, Q, V, d, rpb, out);
    }
    return 0;
}

// =====================================================================================================
// Dense route of the same gradient on the matrix cores (bf16 features, d = 128, Q <= 96, V <= 64: config-2 and anything
// smaller).  Per caption x image pair the terms form a sparse Q x V weight matrix
//     W_ba[q,v] = c1 gV[b,a,q] [v == argV[b,a,q]]  +  c2 gQ[b,a,v] [q == argQ[b,a,v]]
// and  g_txt[b] = sum_a W_ba vis[a],  g_vis[a] = sum_b W_ba^T txt[b].  The sparse kernels above read one 256-byte feature
// row per term from L2 (15.4 M rows, 3.9 GB at config-2: 0.29 + 0.29 ms, L2-bandwidth bound).  Here a pair is 96 + 72
// v_mfma_f32_16x16x32_bf16 whose A operand W (rows of this side, contraction over the other side's positions) is BUILT IN
// REGISTERS from the four small arrays -- two compares and two selects per element -- and whose B operand is the other
// side's feature tile, read once per pair per block through LDS (0.6 GB of L2 -> LDS traffic in all).  The B operand
// must be contraction-major, so the features are transposed once per call into a [tensor][d][Kp] scratch (2.4 + 5.4 MB).
// W is rounded to bf16 (2^-9 relative per term), which is the precision the bf16 path returns its gradients in anyway;
// fp32 features keep the sparse kernels (exact fp32 products).  One wave per 16-row tile, accumulation over the outer
// loop in registers, fixed order: reproducible, no atomics.
// =====================================================================================================
constexpr int kGdD = 128, kGdChunk = 8;   // feature width; pairs per staged chunk of the small arrays

// feat [O][K][128] bf16 -> featT [O][128][Kp] bf16 (zero-padded): block = one 32-row strip of one tensor o, LDS transpose
__global__ __launch_bounds__(256) void ground_transpose_kernel(const uint16_t* __restrict__ feat, int K, int Kp,
                                                               uint16_t* __restrict__ featT) {
    __shared__ uint16_t t[32][kGdD + 2];
    const int o = blockIdx.x, k0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * kGdD; i += 256) {
        const int k = i / kGdD, c = i - k * kGdD;
        t[k][c] = k0 + k < K ? feat[((size_t)o * K + k0 + k) * kGdD + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * kGdD; i += 256) {
        const int c = i >> 5, k = i & 31;
        featT[((size_t)o * kGdD + c) * Kp + k0 + k] = t[k][c];
    }
}

// both sides' transposes in one launch (a ~7 us launch saved per training step): blockIdx.y < nyA -> tensor A, else tensor B
__global__ __launch_bounds__(256) void ground_transpose2_kernel(const uint16_t* __restrict__ featA, int KA, int KpA, uint16_t* __restrict__ outA,
                                                                int nyA, const uint16_t* __restrict__ featB, int KB, int KpB,
                                                                uint16_t* __restrict__ outB) {
    __shared__ uint16_t t[32][kGdD + 2];
    const bool second = (int)blockIdx.y >= nyA;
    const uint16_t* feat = second ? featB : featA;
    uint16_t* featT = second ? outB : outA;
    const int K = second ? KB : KA, Kp = second ? KpB : KpA;
    const int o = blockIdx.x, k0 = ((int)blockIdx.y - (second ? nyA : 0)) * 32;
    for (int i = threadIdx.x; i < 32 * kGdD; i += 256) {
        const int k = i / kGdD, c = i - k * kGdD;
        t[k][c] = k0 + k < K ? feat[((size_t)o * K + k0 + k) * kGdD + c] : (uint16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * kGdD; i += 256) {
        const int c = i >> 5, k = i & 31;
        featT[((size_t)o * kGdD + c) * Kp + k0 + k] = t[k][c];
    }
}

typedef __attribute__((ext_vector_type(8))) __bf16 gd_bf16x8;
typedef __attribute__((ext_vector_type(4))) float gd_f32x4;

// SIDE 0: fix = caption b, rows = its queries (M = Q), outer o = image a, contraction over regions (K = V);
//         self terms = (argV, gV) per row, other terms = (argQ, gQ) per contraction position.
// SIDE 1: fix = image a, rows = its regions (M = V), outer o = caption b, contraction over queries (K = Q); roles swapped.
// NKC = Kp / 32 contraction chunks, MT = ceil(M / 16) row tiles.  Block = 4 waves tiling the [MT x 8] output tiles RW x CW
// (a wave owns RT row tiles x CT column tiles: every A fragment it reads feeds CT MFMAs, every B fragment RT -- one row
// tile per wave, round 2's first version, re-read the whole feature tile in every wave and was LDS-bound).
template <int SIDE, int NKC, int MT, int RW, int SEGL, int PP>
__global__ __launch_bounds__(256) void ground_bwd_dense_kernel(
    const uint16_t* __restrict__ featT, const float* __restrict__ gV, const uint16_t* __restrict__ argV,
    const float* __restrict__ gQ, const uint16_t* __restrict__ argQ, const float* __restrict__ coef, int B, int Q, int V,
    int KT_total, int n_kc, size_t out_split_stride, float* __restrict__ out) {
    // A "pair" below is one STEP UNIT: (outer index o, contraction chunk kc) -- with n_kc = 1 a caption x image pair, with more
    // (the shipped 1369-column layout) one 128-position slice of it; blockIdx.z selects a chunk of MR output rows (image side
    // of that layout: 1369 rows).  Positions outside this block's row chunk / this unit's contraction chunk are simply not
    // elements of its W.  KT_total = row pitch of featT in elements (n_kc * Kp).
    // PP units per step, side by side along the contraction axis: one W [MR][PP * Kp] against one feature tile [d][PP * Kp],
    // so that the per-step latencies (barriers, the dependent LDS chain of the W construction, the drain before the tile goes
    // to LDS, fragment-read latency) are paid once per PP pairs -- the phases of a step do not overlap (ablations).
    // Row pitch in bytes: +32 keeps the ds_read_b128 fragment reads (16 rows x 4 k-groups per instruction, serviced in the
    // hardware's four fixed 16-lane groups, MI355X_MICROARCH.md section LDS) free of bank conflicts; +16 costs 2x on every read
    constexpr int Kp = NKC * 32, KT = PP * Kp, PITCH = KT * 2 + 32;
    constexpr int MR = MT * 16, nthr = 256, CW = 4 / RW, RT = (MT + RW - 1) / RW, CT = 8 / CW;
    const int A = B, M = SIDE == 0 ? Q : V, K = SIDE == 0 ? V : Q;
    // the outer range is split over gridDim.y blocks (occupancy: one block per CU leaves the LDS / barrier latency exposed)
    // (from here on o counts step units: unit u <-> outer index u / n_kc, contraction chunk u % n_kc)
    const int n_units = B * n_kc;
    const int o_per = (n_units + (int)gridDim.y - 1) / (int)gridDim.y, o_begin = blockIdx.y * o_per, O = min(n_units, o_begin + o_per);
    const int m0 = blockIdx.z * MR;                                                           // first output row of this block
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    char* const tile = smem_raw;                                                              // other side's features, [d][KT] bf16
    char* const wt = smem_raw + kGdD * PITCH;                                                 // W [MR][KT] bf16
    // the pairs' four small arrays, a chunk of kGdChunk pairs at a time (the next chunk waits in registers until the last
    // pair of this one has built its W):  [chunk][ argS[MR] | argO[Kp] ] int16  and  [chunk][ valS[MR] | valO[Kp] ] float,
    // the values already scaled by c1 / c2
    constexpr int EW = MR + Kp, CH = kGdChunk;
    static_assert(CH % PP == 0, "a step's pairs come from one chunk");
    int16_t* sarg = reinterpret_cast<int16_t*>(wt + MR * PITCH);
    float* sval = reinterpret_cast<float*>(sarg + CH * EW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fix = blockIdx.x, kg = lane >> 4, ccol = lane & 15;
    const int rt0 = (wave / CW) * RT, ct0 = (wave % CW) * CT;
    const float c_self = SIDE == 0 ? coef[0] : coef[1], c_other = SIDE == 0 ? coef[1] : coef[0];
    const float* g_self = SIDE == 0 ? gV : gQ;
    const uint16_t* a_self = SIDE == 0 ? argV : argQ;
    const float* g_other = SIDE == 0 ? gQ : gV;
    const uint16_t* a_other = SIDE == 0 ? argQ : argV;
    auto pair = [&](int oo) -> size_t { return SIDE == 0 ? (size_t)fix * A + oo : (size_t)oo * A + fix; };
    if (o_begin >= O || m0 >= M) return;

    // W is built in LDS by scatter: thread t < M owns row t (one non-zero per pair: column argself[t]), thread t < K owns
    // column t (one non-zero: row argother[t]).  Where the two kinds meet -- row r points at column t AND column t points at
    // row r -- the column owner writes the sum and the row owner stays out (each sees the other's index in the staged arrays),
    // so no element has two writers; each owner clears what it wrote for the previous step, so the tile is never re-zeroed.
    // (the feature tile is zeroed too: only the 16-byte segments that hold contraction positions < K are ever staged,
    //  the padding up to Kp stays zero)
    for (int i = tid; i < (kGdD + MR) * PITCH / 16; i += nthr) reinterpret_cast<uint4*>(smem_raw)[i] = make_uint4(0, 0, 0, 0);
    constexpr int NV = (kGdD * SEGL + nthr - 1) / nthr;   // 16-byte vectors of one pair's feature tile per thread
    constexpr int segs = SEGL, nvec = kGdD * SEGL;         // segments per row that hold data (SEGL >= ceil(K / 8))
    // One register set holds the NEXT step's feature tiles in flight: loaded (branch-free) right after the previous contents
    // went to LDS, consumed one whole step later, and nothing younger is outstanding at that point, so the drain the compiler
    // puts there (s_waitcnt vmcnt(0): it does not count loads across the loop's back edge) costs nothing extra.
    uint4 xs[PP][NV];
#pragma unroll
    for (int q = 0; q < PP; ++q)
#pragma unroll
        for (int j = 0; j < NV; ++j) xs[q][j] = make_uint4(0, 0, 0, 0);   // (an array first written under a condition ends up in scratch)
    auto stage_load = [&](int o, uint4 (*xs)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PP; ++q) {
            const int u = min(o + q, O - 1), oo = u / n_kc, kc = u - oo * n_kc;
            const uint4* src = reinterpret_cast<const uint4*>(featT + ((size_t)oo * kGdD * KT_total + (size_t)kc * Kp));
            const int rowv = KT_total >> 3;   // uint4 per feature row
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = min(tid + j * nthr, nvec - 1);
                const int row = i / segs, seg = i - row * segs;
                xs[q][j] = src[row * rowv + seg];
            }
        }
    };
    auto stage_tile = [&](const uint4 (*xs)[NV]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PP; ++q)
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = tid + j * nthr;
                if (i < nvec) {
                    const int row = i / segs, seg = i - row * segs;
                    *reinterpret_cast<uint4*>(tile + row * PITCH + q * Kp * 2 + seg * 16) = xs[q][j];
                }
            }
    };
    // The small arrays come from HBM (each is read exactly once): a whole chunk of pairs is fetched with one burst of
    // independent loads (raw values only -- arithmetic here would make the wave wait for what it has just issued) while the
    // previous chunk is being worked on, so that latency is paid once per kGdChunk pairs instead of once per pair.
    constexpr int NE = (CH * EW + nthr - 1) / nthr;   // (argument, value) entries per thread and chunk
    int carg[NE];
    float cval[NE];
#pragma unroll
    for (int j = 0; j < NE; ++j) { carg[j] = 0; cval[j] = 0.f; }
    // (the 16-bit positions are fetched as the aligned dword that holds them and picked apart when they go to LDS: a 16-bit
    //  load is widened by an instruction of its own, which the compiler places right behind the burst -- a drain)
    unsigned cpar = 0;   // bit j: entry j's position sits in the upper half of its dword
    // (by-value captures: a `cond ? a : b` over two by-reference captures is a run-time index into the closure object, which
    //  then stays in memory -- scratch -- together with everything it refers to)
    auto chunk_load = [=](int oc, int* carg, float* cval, unsigned& cpar) __attribute__((always_inline)) {
        cpar = 0;
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int id = min(tid + j * nthr, CH * EW - 1), p = id / EW, e = id - p * EW;
            const int u = min(oc + p, O - 1), oo = u / n_kc, k0 = (u - oo * n_kc) * Kp;
            const size_t pr = pair(oo);
            const bool other = e >= MR;
            const size_t at = other ? pr * K + min(k0 + e - MR, K - 1) : pr * M + min(m0 + e, M - 1);
            const uint32_t* ap = reinterpret_cast<const uint32_t*>(other ? a_other : a_self) + (at >> 1);
            const float* gp = (other ? g_other : g_self) + at;
            carg[j] = (int)*ap;
            cval[j] = *gp;
            cpar |= ((unsigned)at & 1u) << j;
        }
    };
    auto chunk_store = [=](int oc, const int* carg, const float* cval, unsigned cpar) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int id = tid + j * nthr;
            if (id < CH * EW) {
                const int p = id / EW, e = id - p * EW;
                const int u = min(oc + p, O - 1), k0 = (u % n_kc) * Kp;
                const bool other = e >= MR;
                // position relative to this block's row chunk / this unit's contraction chunk, -1 when it lies outside
                const int av = ((carg[j] >> (((cpar >> j) & 1u) << 4)) & 0xffff) - (other ? m0 : k0);
                const bool ok = other ? (k0 + e - MR < K && av >= 0 && av < MR && m0 + av < M)
                                      : (m0 + e < M && av >= 0 && av < Kp && k0 + av < K);
                sarg[id] = (int16_t)(ok ? av : -1);
                // (bit select, not `other ? c_other : c_self`: the compiler turns that into a two-entry table in scratch)
                const unsigned pick = other ? ~0u : 0u;
                sval[id] = __uint_as_float((__float_as_uint(c_other) & pick) | (__float_as_uint(c_self) & ~pick)) * cval[j];
            }
        }
    };
    auto f2bf = [](float f) -> uint16_t { return __builtin_bit_cast(uint16_t, (__bf16)f); };
    gd_f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = gd_f32x4{0.f, 0.f, 0.f, 0.f};
    int ps_old[PP], po_old[PP];
#pragma unroll
    for (int q = 0; q < PP; ++q) ps_old[q] = po_old[q] = -1;
    chunk_load(o_begin, carg, cval, cpar);
    stage_load(o_begin, xs);
    __syncthreads();                       // the zero fill above
    chunk_store(o_begin, carg, cval, cpar);
    stage_tile(xs);
    stage_load(min(o_begin + PP, O - 1), xs);
    __syncthreads();
    for (int o = o_begin; o < O; o += PP) {
        const int cp = (o - o_begin) % CH;
        if (cp == 0 && o + CH < O) chunk_load(o + CH, carg, cval, cpar);   // the chunk after this one: in flight during this step
        // ---- W of this step: clear last step's elements, write this step's (same owner thread: program order) ----
        auto welem = [&](int row, int col) { return reinterpret_cast<uint16_t*>(wt + row * PITCH + col * 2); };
#ifndef VLG_GD_NOW
        {
            // every read of the staged arrays first (for all PP pairs: independent chains), then the clears and writes.  BRANCH-FREE
            // (round 3): unconditional LDS reads at clamped indices, integer flags, writes that go to the row's pitch padding when
            // they are not wanted -- as `a && b[i]` / `if (c) *p = v` every read became an exec-masked block with its own
            // s_waitcnt lgkmcnt(0), a dozen LDS round trips in a row per pair (see ground_bwd_ws_kernel).
            const int prow = min(tid, MR - 1), pcol = min(tid, Kp - 1);
            uint16_t* const dummy = welem(prow, KT);   // this row's padding: never read
            int a_ps[PP], a_po[PP], s_raw[PP], o_raw[PP];
            float vrow[PP], vcolb[PP];
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const int16_t* argS = sarg + (cp + q) * EW;
                const int16_t* argO = argS + MR;
                const float* valS = sval + (cp + q) * EW;
                const float* valO = valS + MR;
                a_ps[q] = argO[max(ps_old[q], 0)];
                a_po[q] = argS[max(po_old[q], 0)];
                s_raw[q] = argS[prow];
                o_raw[q] = argO[pcol];
                vrow[q] = valS[prow];
                vcolb[q] = valO[pcol];
            }
            int my_s[PP], my_o[PP], a_ms[PP], a_mo[PP];
            float v_mo[PP];
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const int live = o + q < O ? -1 : 0;             // the last step of an odd range has one pair: its second half only clears
                const int16_t* argS = sarg + (cp + q) * EW;
                const int16_t* argO = argS + MR;
                const float* valS = sval + (cp + q) * EW;
                my_s[q] = (live & (tid < MR ? -1 : 0)) ? s_raw[q] : -1;   // already -1 where invalid
                my_o[q] = (live & (tid < Kp ? -1 : 0)) ? o_raw[q] : -1;
                a_ms[q] = argO[max(my_s[q], 0)];
                a_mo[q] = argS[max(my_o[q], 0)];
                v_mo[q] = valS[max(my_o[q], 0)];
            }
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const int live = o + q < O ? 1 : 0;
                // (an old element is left alone when ANOTHER thread writes this step's value to the same place -- the column
                //  owner of my old column pointing at my row, or the row owner of my old row pointing at my column -- since its
                //  write and my clear are unordered; my own clear-then-write is program order, a wave's LDS operations complete
                //  in order)
                const int clr_s = (ps_old[q] >= 0 ? 1 : 0) & (1 ^ (live & (a_ps[q] == tid ? 1 : 0)));
                const int clr_o = (po_old[q] >= 0 ? 1 : 0) & (1 ^ (live & (a_po[q] == tid ? 1 : 0)));
                const int row_w = (my_s[q] >= 0 ? 1 : 0) & (a_ms[q] != tid ? 1 : 0);   // else the column owner writes the sum
                const int col_w = my_o[q] >= 0 ? 1 : 0;
                const float vcol = vcolb[q] + (a_mo[q] == tid ? v_mo[q] : 0.f);
                uint16_t* p0 = clr_s ? welem(prow, q * Kp + max(ps_old[q], 0)) : dummy;
                uint16_t* p1 = clr_o ? welem(max(po_old[q], 0), q * Kp + pcol) : dummy;
                uint16_t* p2 = row_w ? welem(prow, q * Kp + max(my_s[q], 0)) : dummy;
                uint16_t* p3 = col_w ? welem(max(my_o[q], 0), q * Kp + pcol) : dummy;
                *p0 = 0;
                *p1 = 0;
                *p2 = f2bf(vrow[q]);
                *p3 = f2bf(vcol);
                ps_old[q] = row_w ? my_s[q] : -1;
                po_old[q] = col_w ? my_o[q] : -1;
            }
        }
#endif
        __syncthreads();                                          // W and the tile of this step complete
        // Fragment reads are issued in batches ahead of their MFMAs (sched_barrier pins them there): left to itself hipcc
        // reuses ONE register quad for the B fragments -- read, wait lgkmcnt(0), MFMAs, next read -- which exposes a full
        // LDS latency per fragment.
#ifndef VLG_GD_NOMFMA
        {
            constexpr int NK = PP * NKC;
            gd_bf16x8 af[2][RT], bf[2][CT];   // two fragment sets: chunk kc+1 is read while chunk kc's MFMAs run
            auto frags = [&](int kc, gd_bf16x8* a, gd_bf16x8* b) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < RT; ++r)
                    a[r] = *reinterpret_cast<const gd_bf16x8*>(wt + (min(rt0 + r, MT - 1) * 16 + ccol) * PITCH + (kc * 4 + kg) * 16);
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    b[c] = *reinterpret_cast<const gd_bf16x8*>(tile + ((ct0 + c) * 16 + ccol) * PITCH + (kc * 4 + kg) * 16);
            };
            frags(0, af[0], bf[0]);
#pragma unroll
            for (int kc = 0; kc < NK; ++kc) {
                if (kc + 1 < NK) frags(kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < CT; ++c)
                        acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kc & 1][r], bf[kc & 1][c], acc[r][c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#endif
        __syncthreads();                                          // every wave is done with the tile and W before they change
#ifndef VLG_GD_NOSTAGE
        stage_tile(xs);                                           // the next step's pairs
        if (cp + PP == CH && o + PP < O) chunk_store(o + PP, carg, cval, cpar);   // this chunk's last W is built: its arrays make way
        stage_load(min(o + 2 * PP, O - 1), xs);
#endif
    }
    // accumulator tile: lane l, register n <-> row 4 (l >> 4) + n, column l & 15
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (rt0 + r >= MT) break;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rr = m0 + (rt0 + r) * 16 + kg * 4 + n;
            if (rr < M) {
                // out_split_stride != 0: every split writes its own partial sum (ground_partial_sum_kernel adds them in a
                // fixed order); else two partial sums meet in a zeroed output by atomics, which is order-free for two addends
                float* dst = out + (size_t)blockIdx.y * out_split_stride + ((size_t)fix * M + rr) * kGdD + ct0 * 16 + ccol;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (gridDim.y > 1 && out_split_stride == 0) atomicAdd(dst + c * 16, acc[r][c][n]);
                    else dst[c * 16] = acc[r][c][n];
                }
            }
        }
    }
}

// =====================================================================================================
// The same contraction with the two halves of a step on DIFFERENT wavefronts (round 3; one contraction chunk per pair, i.e.
// config-2's widths).  In the kernel above a step is: build W (a dependent chain of LDS reads and scattered writes) | barrier |
// fragment reads + MFMAs | barrier | next feature tile to LDS -- and nothing of one phase overlaps the next; with two blocks per CU
// the matrix cores are busy a quarter of the time (164 + 178 us per call at config-2 against ~45 us of MFMA work each).
// Here a block is 12 waves, one block per CU: waves 4-11 PRODUCE step t + 1 into the other LDS buffer -- two waves build W, four
// stage the feature tile, two stream the pairs' small arrays, each with the code of the kernel above -- while waves 0-3 CONSUME
// step t (fragment reads + MFMAs, the same [MT x 8] tiling), ONE barrier per step.  Both W and the feature tile are double-buffered; so are the staged small arrays (the producers would
// otherwise need a barrier of their own between the last read of a chunk and the store of the next one).  No split of the outer
// range by default: a block sweeps all partners of its caption / image, the result is written once -- no atomics, no zero fill.
// =====================================================================================================
#ifdef VLG_GW_STAMP   // tools/ experiment: per-role cycles from a barrier's release to the arrival at the next one (printed for two blocks)
#define GW_STAMP_DECL unsigned long long st_busy = 0, st_t = __builtin_amdgcn_s_memtime();
#define GW_SYNC() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); st_busy += __builtin_amdgcn_s_memtime() - st_t; __syncthreads(); st_t = __builtin_amdgcn_s_memtime(); } while (0)
#define GW_STAMP_OUT(role) do { if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == 100)) printf("block %d wave %d role %s: %llu cycles busy over %d steps = %llu per step\n", (int)blockIdx.x, wave, role, st_busy, T, st_busy / (unsigned long long)T); } while (0)
#else
#define GW_STAMP_DECL
#define GW_SYNC() __syncthreads()
#define GW_STAMP_OUT(role)
#endif
constexpr int kGwThreads = 768;   // twelve waves: 4 consumers, 2 builders, 4 tile stagers, 2 array stagers
// KS = 2: the consumers split the contraction axis between two wave pairs (each wave twice the column tiles, half the chunks:
// fewer fragment reads per MFMA -- the image side's 3 x 2 tiles per wave read 5 fragments per 6 MFMAs and the LDS, not the matrix
// cores, set its step); the upper pair hands its accumulators over through the LDS once, at the end.
template <int SIDE, int NKC, int MT, int RW, int SEGL, int PP, int UW, int KS>
__global__ __launch_bounds__(kGwThreads) void ground_bwd_ws_kernel(
    const uint16_t* __restrict__ featT, const float* __restrict__ gV, const uint16_t* __restrict__ argV,
    const float* __restrict__ gQ, const uint16_t* __restrict__ argQ, const float* __restrict__ coef, int B, int Q, int V,
    float* __restrict__ out) {
    // UW = unit width: the columns a pair occupies in the step's contraction axis (Kp of featT, or less when the positions
    // past K are not staged -- config-2's 36 regions: 40 instead of 64, three pairs per 128 columns instead of two)
    constexpr int Kp = NKC * 32, KT = (PP * UW + 31) / 32 * 32, PITCH = KT * 2 + 32, NK = KT / 32;
    constexpr int MR = MT * 16, CW = 4 / RW / KS, RT = (MT + RW - 1) / RW, CT = 8 / CW;
    static_assert(KS == 1 || (KS == 2 && NK % 2 == 0 && 4 % (RW * KS) == 0), "two halves of the chunks");
    static_assert(UW % 8 == 0 && UW <= Kp && SEGL * 8 <= UW, "a unit is whole 16-byte segments");
    static_assert(MR <= 128 && Kp <= 128, "row and column owners are the 128 threads of the two builder waves");
    const int A = B, M = SIDE == 0 ? Q : V, K = SIDE == 0 ? V : Q;
    const int o_per = (B + (int)gridDim.y - 1) / (int)gridDim.y, o_begin = blockIdx.y * o_per, O = min(B, o_begin + o_per);
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int TILE_B = kGdD * PITCH, W_B = MR * PITCH;
    char* const tile0 = smem_raw;                       // [2][d][KT] bf16
    char* const wt0 = smem_raw + 2 * TILE_B;            // [2][MR][KT] bf16
    constexpr int EW = MR + Kp;
    constexpr int CHP = PP == 3 ? 6 : kGdChunk;         // pairs per chunk of the small arrays: a multiple of PP
    constexpr int SPC = CHP / PP;                       // steps per chunk
    static_assert(CHP % PP == 0 && SPC >= 2, "a chunk in flight needs more than one step to land");
    int16_t* const sarg0 = reinterpret_cast<int16_t*>(wt0 + 2 * W_B);          // [2][CHP][EW]
    float* const sval0 = reinterpret_cast<float*>(sarg0 + 2 * CHP * EW);       // [2][CHP][EW]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fix = blockIdx.x;
    if (o_begin >= O) return;
    const int T = (O - o_begin + PP - 1) / PP;          // steps
    for (int i = tid; i < (2 * TILE_B + 2 * W_B) / 16; i += kGwThreads) reinterpret_cast<uint4*>(smem_raw)[i] = make_uint4(0, 0, 0, 0);
    // Every role runs the same barrier sequence: three in the prologue (zero fill | chunk 0 of the small arrays | step 0 built and
    // staged), then one per step.  A role's vector-memory loads are its own wave's: an s_waitcnt vmcnt(0) (all hipcc can place
    // across a loop's back edge) in the wave that streams the small arrays from HBM does not hold up the wave that stages tiles.

    if (wave >= 10) {
        // ---- waves 10, 11: the pairs' small arrays, HBM -> registers one chunk ahead -> LDS (the half the builders are not reading) ----
        const int atid = tid - 640;
        constexpr int nthr = 128, NE = (CHP * EW + nthr - 1) / nthr;
        const float c_self = SIDE == 0 ? coef[0] : coef[1], c_other = SIDE == 0 ? coef[1] : coef[0];
        const float* g_self = SIDE == 0 ? gV : gQ;
        const uint16_t* a_self = SIDE == 0 ? argV : argQ;
        const float* g_other = SIDE == 0 ? gQ : gV;
        const uint16_t* a_other = SIDE == 0 ? argQ : argV;
        auto pair = [&](int oo) -> size_t { return SIDE == 0 ? (size_t)fix * A + oo : (size_t)oo * A + fix; };
        int carg[NE];
        float cval[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) { carg[j] = 0; cval[j] = 0.f; }
        unsigned cpar = 0;
        auto chunk_load = [=](int oc, int* carg, float* cval, unsigned& cpar) __attribute__((always_inline)) {
            cpar = 0;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int id = min(atid + j * nthr, CHP * EW - 1), p = id / EW, e = id - p * EW;
                const int oo = min(oc + p, O - 1);
                const size_t pr = pair(oo);
                const bool other = e >= MR;
                const size_t at = other ? pr * K + min(e - MR, K - 1) : pr * M + min(e, M - 1);
                const uint32_t* ap = reinterpret_cast<const uint32_t*>(other ? a_other : a_self) + (at >> 1);
                const float* gp = (other ? g_other : g_self) + at;
                carg[j] = (int)*ap;
                cval[j] = *gp;
                cpar |= ((unsigned)at & 1u) << j;
            }
        };
        auto chunk_store = [=](int half, const int* carg, const float* cval, unsigned cpar) __attribute__((always_inline)) {
            int16_t* sarg = sarg0 + half * CHP * EW;
            float* sval = sval0 + half * CHP * EW;
#pragma unroll
            for (int j = 0; j < NE; ++j) {
                const int id = atid + j * nthr;
                if (id < CHP * EW) {
                    const int p = id / EW, e = id - p * EW;
                    const bool other = e >= MR;
                    const int av = (carg[j] >> (((cpar >> j) & 1u) << 4)) & 0xffff;
                    const bool ok = other ? (e - MR < K && av < M) : (e < M && av < K);
                    sarg[id] = (int16_t)(ok ? av : -1);
                    const unsigned pick = other ? ~0u : 0u;
                    sval[id] = __uint_as_float((__float_as_uint(c_other) & pick) | (__float_as_uint(c_self) & ~pick)) * cval[j];
                }
            }
        };
        chunk_load(o_begin, carg, cval, cpar);
        __syncthreads();                                 // (the zero fill)
        chunk_store(0, carg, cval, cpar);
        if (SPC < T) chunk_load(o_begin + CHP, carg, cval, cpar);
        __syncthreads();                                 // (chunk 0 visible to the builders)
        __syncthreads();
        GW_STAMP_DECL
        for (int t = 0; t < T; ++t) {
            const int u = t + 1;
            // the chunk that starts with the NEXT step: its half was last read a whole chunk ago (the current chunk sits in the
            // other half), and the barrier below publishes it
#ifndef VLG_GW_NOARGS   // tools/ ablations (results are wrong with any of these)
            if (u < T && (u + 1) % SPC == 0 && (u + 1) < T) {
                const int c = (u + 1) / SPC;
                chunk_store(c & 1, carg, cval, cpar);
                if ((c + 1) * SPC < T) chunk_load(o_begin + (c + 1) * CHP, carg, cval, cpar);
            }
#endif
            GW_SYNC();
        }
        GW_STAMP_OUT("arrays");
        return;
    }

    if (wave >= 6) {
        // ---- waves 6-9: the other side's feature tiles, L2 -> registers one step ahead -> LDS (the buffer the consumers are not reading) ----
        const int ttid = tid - 384;
        constexpr int nthr = 256, NV = (kGdD * SEGL + nthr - 1) / nthr, segs = SEGL, nvec = kGdD * SEGL;
        // the register set of the tile in flight, addressed by a compile-time set number.  (Round 6 measured TWO sets -- step u + 2
        // requested when step u is written, counted vmcnt waits: 131.4 / 102.0 us against 132.0 / 101.5: a tile's reads are back within a
        // step.  Arrays handed to the lambdas by pointer or reference went to scratch memory once there were two of them.)
        uint4 xs[1][PP][NV];
#pragma unroll
        for (int q = 0; q < PP; ++q)
#pragma unroll
            for (int j = 0; j < NV; ++j) xs[0][q][j] = make_uint4(0, 0, 0, 0);
        // a thread's source offsets inside a pair's [d][Kp] block never change: one buffer descriptor over the feature tensor, the
        // per-thread byte offsets in NV registers and the pair's base as the scalar offset -- a read is one `buffer_load_dwordx4` with no
        // address arithmetic in the loop.  (The 64-bit per-thread addresses the compiler built every step reused registers of the step
        // before, and the hazard it then assumed put an s_waitcnt vmcnt(2..4) between the reads of ONE step: every step waited for
        // reads it had issued a few instructions earlier.)
        typedef unsigned int gw_u32x4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t feat_rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)featT, 0, A * kGdD * Kp * 2, 0x00020000);
        int soff[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = min(ttid + j * nthr, nvec - 1);
            const int row = i / segs, seg = i - row * segs;
            soff[j] = (row * (Kp >> 3) + seg) * 16;
        }
        auto stage_load = [&](int o, auto set) __attribute__((always_inline)) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const int oo = min(o + q, O - 1);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const gw_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(feat_rs, soff[j], oo * (kGdD * Kp * 2), 0);
                    xs[S][q][j] = make_uint4(v[0], v[1], v[2], v[3]);
                }
            }
        };
        auto stage_tile = [&](char* tile, auto set) __attribute__((always_inline)) {
            constexpr int S = decltype(set)::value;
#pragma unroll
            for (int q = 0; q < PP; ++q)
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = ttid + j * nthr;
                    if (i < nvec) {
                        const int row = i / segs, seg = i - row * segs;
                        *reinterpret_cast<uint4*>(tile + row * PITCH + q * UW * 2 + seg * 16) = xs[S][q][j];
                    }
                }
        };
        constexpr std::integral_constant<int, 0> s0{};
        stage_load(o_begin, s0);
        __syncthreads();                                 // (the zero fill)
        __syncthreads();
        stage_tile(tile0, s0);
        stage_load(o_begin + PP, s0);
        __syncthreads();
        GW_STAMP_DECL
        for (int t = 0; t < T; ++t) {
            const int u = t + 1;
#ifndef VLG_GW_NOSTAGE
            if (u < T) {
                stage_tile(tile0 + (u & 1) * TILE_B, s0);
                stage_load(o_begin + (u + 1) * PP, s0);
            }
#endif
            GW_SYNC();
        }
        GW_STAMP_OUT("tiles");
        return;
    }

    if (wave >= 4) {
        // ---- waves 4, 5: W of step t + 1, built in LDS by scatter (the scheme of the kernel above: thread r < M owns row r, thread
        //      k < K owns column k, the column owner writes the sum where the two meet, every owner clears what it wrote into this
        //      buffer two steps ago); LDS traffic only ----
        const int ptid = tid - 256;
        auto f2bf = [](float f) -> uint16_t { return __builtin_bit_cast(uint16_t, (__bf16)f); };
        int ps_old[2][PP], po_old[2][PP];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < PP; ++q) ps_old[h][q] = po_old[h][q] = -1;
        // BRANCH-FREE: every LDS read is unconditional (clamped index, the flags are integer ands) and every write goes either to its
        // element or to the row's pitch padding -- written as `a && b[i]` / `if (c) *p = v`, hipcc emits one exec-masked block per
        // read, each with its own s_waitcnt lgkmcnt(0): 36 LDS round trips in a row per step (measured: the builders alone took
        // 2500 cycles per step, three times the consumers' MFMAs).
        auto build = [&](int u, char* wt, int* ps_o, int* po_o) __attribute__((always_inline)) {
            const int o = o_begin + u * PP, cp = (u % SPC) * PP, half = (u / SPC) & 1;
            const int prow = min(ptid, MR - 1), pcol = min(ptid, Kp - 1);
            auto welem = [&](int row, int col) { return reinterpret_cast<uint16_t*>(wt + row * PITCH + col * 2); };
            uint16_t* const dummy = welem(prow, KT);   // this row's padding: never read
            int a_ps[PP], a_po[PP], s_raw[PP], o_raw[PP];
            float vrow[PP], vcolb[PP];
#pragma unroll
            for (int q = 0; q < PP; ++q) {   // level 1: everything addressed by known indices
                const int16_t* argS = sarg0 + (half * CHP + cp + q) * EW;
                const int16_t* argO = argS + MR;
                const float* valS = sval0 + (half * CHP + cp + q) * EW;
                const float* valO = valS + MR;
                a_ps[q] = argO[max(ps_o[q], 0)];
                a_po[q] = argS[max(po_o[q], 0)];
                s_raw[q] = argS[prow];
                o_raw[q] = argO[pcol];
                vrow[q] = valS[prow];
                vcolb[q] = valO[pcol];
            }
            int my_s[PP], my_o[PP], a_ms[PP], a_mo[PP];
            float v_mo[PP];
#pragma unroll
            for (int q = 0; q < PP; ++q) {   // level 2: addressed by what level 1 returned
                const int live = o + q < O ? -1 : 0;
                const int16_t* argS = sarg0 + (half * CHP + cp + q) * EW;
                const int16_t* argO = argS + MR;
                const float* valS = sval0 + (half * CHP + cp + q) * EW;
                my_s[q] = (live & (ptid < MR ? -1 : 0)) ? s_raw[q] : -1;
                my_o[q] = (live & (ptid < Kp ? -1 : 0)) ? o_raw[q] : -1;
                a_ms[q] = argO[max(my_s[q], 0)];
                a_mo[q] = argS[max(my_o[q], 0)];
                v_mo[q] = valS[max(my_o[q], 0)];
            }
#pragma unroll
            for (int q = 0; q < PP; ++q) {
                const int live = o + q < O ? 1 : 0;
                // (an old element is left alone when ANOTHER thread writes this step's value to the same place -- the column owner of
                //  my old column pointing at my row, or the row owner of my old row pointing at my column -- since its write and my
                //  clear are unordered; my own clear-then-write is program order, a wave's LDS operations complete in order)
                const int clr_s = (ps_o[q] >= 0 ? 1 : 0) & (1 ^ (live & (a_ps[q] == ptid ? 1 : 0)));
                const int clr_o = (po_o[q] >= 0 ? 1 : 0) & (1 ^ (live & (a_po[q] == ptid ? 1 : 0)));
                const int row_w = (my_s[q] >= 0 ? 1 : 0) & (a_ms[q] != ptid ? 1 : 0);   // else the column owner writes the sum
                const int col_w = my_o[q] >= 0 ? 1 : 0;
                const float vcol = vcolb[q] + (a_mo[q] == ptid ? v_mo[q] : 0.f);
                uint16_t* p0 = clr_s ? welem(prow, q * UW + max(ps_o[q], 0)) : dummy;
                uint16_t* p1 = clr_o ? welem(max(po_o[q], 0), q * UW + pcol) : dummy;
                uint16_t* p2 = row_w ? welem(prow, q * UW + max(my_s[q], 0)) : dummy;
                uint16_t* p3 = col_w ? welem(max(my_o[q], 0), q * UW + pcol) : dummy;
                *p0 = 0;
                *p1 = 0;
                *p2 = f2bf(vrow[q]);
                *p3 = f2bf(vcol);
                ps_o[q] = row_w ? my_s[q] : -1;
                po_o[q] = col_w ? my_o[q] : -1;
            }
        };
        __syncthreads();                                 // (the zero fill)
        __syncthreads();                                 // (chunk 0 of the small arrays)
        build(0, wt0, ps_old[0], po_old[0]);
        __syncthreads();
        GW_STAMP_DECL
        for (int t = 0; t < T; ++t) {
            const int u = t + 1;
#ifndef VLG_GW_NOBUILD
            if (u < T) {
                if (u & 1) build(u, wt0 + W_B, ps_old[1], po_old[1]);
                else build(u, wt0, ps_old[0], po_old[0]);
            }
#endif
            GW_SYNC();
        }
        GW_STAMP_OUT("build");
        return;
    }

    // ---- waves 0-3: fragment reads + MFMAs of step t, tiling the [MT x 8] output tiles RW x CW ----
    const int kg = lane >> 4, ccol = lane & 15;
    const int ks = wave / (RW * CW), w2 = wave % (RW * CW);
    const int rt0 = (w2 / CW) * RT, ct0 = (w2 % CW) * CT;
    constexpr int NKs = NK / KS;
    gd_f32x4 acc[RT][CT];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[r][c] = gd_f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    __syncthreads();
    __syncthreads();
    GW_STAMP_DECL
    for (int t = 0; t < T; ++t) {
        const char* wt = wt0 + (t & 1) * W_B;
        const char* tile = tile0 + (t & 1) * TILE_B;
        gd_bf16x8 af[2][RT], bf[2][CT];   // two fragment sets: chunk kc+1 is read while chunk kc's MFMAs run
        auto frags = [&](int kc, gd_bf16x8* a, gd_bf16x8* b) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
                a[r] = *reinterpret_cast<const gd_bf16x8*>(wt + (min(rt0 + r, MT - 1) * 16 + ccol) * PITCH + (kc * 4 + kg) * 16);
#pragma unroll
            for (int c = 0; c < CT; ++c)
                b[c] = *reinterpret_cast<const gd_bf16x8*>(tile + ((ct0 + c) * 16 + ccol) * PITCH + (kc * 4 + kg) * 16);
        };
#ifndef VLG_GW_NOMFMA
        frags(ks * NKs, af[0], bf[0]);
#pragma unroll
        for (int kc = 0; kc < NKs; ++kc) {
            if (kc + 1 < NKs) frags(ks * NKs + kc + 1, af[(kc + 1) & 1], bf[(kc + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c)
                    acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kc & 1][r], bf[kc & 1][c], acc[r][c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        GW_SYNC();
    }
    GW_STAMP_OUT("mfma");
    if (KS == 2) {   // (the producers have left: a barrier counts the waves that are still running)
        float* xch = reinterpret_cast<float*>(smem_raw) + (size_t)w2 * (RT * CT * 4 * 64);
        if (ks == 1) {
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int c = 0; c < CT; ++c)
#pragma unroll
                    for (int n = 0; n < 4; ++n) xch[((r * CT + c) * 4 + n) * 64 + lane] = acc[r][c][n];
        }
        __syncthreads();
        if (ks == 1) return;
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[r][c][n] += xch[((r * CT + c) * 4 + n) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        if (rt0 + r >= MT) break;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int rr = (rt0 + r) * 16 + kg * 4 + n;
            if (rr < M) {
                float* dst = out + ((size_t)fix * M + rr) * kGdD + ct0 * 16 + ccol;
#pragma unroll
                for (int c = 0; c < CT; ++c) {
                    if (gridDim.y > 1) atomicAdd(dst + c * 16, acc[r][c][n]);
                    else dst[c * 16] = acc[r][c][n];
                }
            }
        }
    }
}

// splits of the caption side's step units over blockIdx.y for the wide layouts: enough blocks to cover the chip
int ground_dense_split_txt(int B, int V) {
    const int n_kc = (V + 127) / 128, n = n_kc * B;
    int split = std::max(1, std::min(std::min(16, n / 8), (512 + B - 1) / B));
    while (split > 1 && (split - 1) * ((n + split - 1) / split) >= n) --split;   // no empty share: every split writes its partial sum
    return split;
}

// out[i] = part[0][i] + part[1][i] + ... in that order (the deeper splits of the wide layouts)
__global__ __launch_bounds__(256) void ground_partial_sum_kernel(const float* __restrict__ part, int nsplit, size_t n4,
                                                                 float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = reinterpret_cast<const float4*>(part)[i];
    for (int k = 1; k < nsplit; ++k) {
        const float4 b = reinterpret_cast<const float4*>(part)[(size_t)k * n4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    reinterpret_cast<float4*>(out)[i] = a;
}

static int launch_bwd_dense(const void* txt, const void* vis, const float* gV, const uint16_t* argV, const float* gQ,
                            const uint16_t* argQ, const float* coef, int B, int Q, int V, uint16_t* scratch, float* partial,
                            float* g_txt, float* g_vis, hipStream_t s) {
    // config-2 widths (V <= 64): one contraction chunk per pair, two pairs per step.  Wider region axes (the shipped layout's
    // 1369 columns): caption side -- 128-position contraction chunks, one per step; image side -- 96-row output chunks in
    // blockIdx.z, contraction over the <= 96 queries.
    const bool wide = V > 64;
    const int KpQ = 96, KtV = wide ? (V + 127) / 128 * 128 : 64;
    uint16_t* visT = scratch;                              // [B][128][KtV]
    uint16_t* txtT = scratch + (size_t)B * kGdD * KtV;     // [B][128][96]
    auto lds = [](int Kp, int mt, int pp) {
        return (size_t)(kGdD + mt * 16) * (pp * Kp * 2 + 32) + (size_t)kGdChunk * (mt * 16 + Kp) * (2 + 4);
    };
    // config-2's widths: producer / consumer wavefronts, one block per caption / image, written once (ground_bwd_ws_kernel)
    const bool ws_ok = !wide && !VLG_ENV("VLG_GD_OLD") && (size_t)B * kGdD * 128 * 2 < ((size_t)1 << 31);   // (32-bit buffer offsets of the tile reads)
    auto lds_ws = [](int Kp, int mt, int pp, int uw) {
        const int kt = (pp * uw + 31) / 32 * 32, chp = pp == 3 ? 6 : kGdChunk;
        return 2 * (size_t)(kGdD + mt * 16) * (kt * 2 + 32) + 2 * (size_t)chp * (mt * 16 + Kp) * (2 + 4);
    };
#define VLG_WS(SIDEV, NKCV, MTV, RWV, SEGV, PPV, UWV, FT, OUT)                                                          \
    do {                                                                                                               \
        auto kern = ground_bwd_ws_kernel<SIDEV, NKCV, MTV, RWV, SEGV, PPV, UWV, (SIDEV == 1 && RWV == 1 ? 2 : 1)>;        \
        const size_t nb = lds_ws(NKCV * 32, MTV, PPV, UWV);                                                            \
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb);    \
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));                \
        hipLaunchKernelGGL(kern, dim3(B, 1), dim3(kGwThreads), nb, s, FT, gV, argV, gQ, argQ, coef, B, Q, V, OUT);             \
    } while (0)
#define VLG_WS0(MTV, RWV)                                                                                              \
    do { if (V <= 40) VLG_WS(0, 2, MTV, RWV, 5, 3, 40, visT, g_txt); else VLG_WS(0, 2, MTV, RWV, 8, 2, 64, visT, g_txt); } while (0)
    const bool both_ws = g_txt && g_vis && ws_ok && V <= 48;
    if (both_ws) {
        hipLaunchKernelGGL(ground_transpose2_kernel, dim3(B, KtV / 32 + KpQ / 32), dim3(256), 0, s, (const uint16_t*)vis, V, KtV, visT, KtV / 32,
                           (const uint16_t*)txt, Q, KpQ, txtT);
        if (int rc = check_launch("ground_transpose2_kernel")) return rc;
    }
    if (g_txt && ws_ok) {
        if (!both_ws) hipLaunchKernelGGL(ground_transpose_kernel, dim3(B, KtV / 32), dim3(256), 0, s, (const uint16_t*)vis, V, KtV, visT);
        switch ((Q + 15) / 16) {
            case 1: VLG_WS0(1, 1); break;
            case 2: VLG_WS0(2, 2); break;
            case 3: VLG_WS0(3, 1); break;
            case 4: VLG_WS0(4, 2); break;
            case 5: VLG_WS0(5, 2); break;
            default: VLG_WS0(6, 2); break;
        }
        if (int rc = check_launch("ground_bwd_ws_kernel")) return rc;
        g_txt = nullptr;
    }
    if (g_vis && ws_ok && V <= 48) {   // (four region tiles: the double buffers do not fit the LDS)
        if (!both_ws) hipLaunchKernelGGL(ground_transpose_kernel, dim3(B, KpQ / 32), dim3(256), 0, s, (const uint16_t*)txt, Q, KpQ, txtT);
        switch ((V + 15) / 16) {
            case 1: VLG_WS(1, 3, 1, 1, 12, 2, 96, txtT, g_vis); break;
            case 2: VLG_WS(1, 3, 2, 2, 12, 2, 96, txtT, g_vis); break;
            default: VLG_WS(1, 3, 3, 1, 12, 2, 96, txtT, g_vis); break;
        }
        if (int rc = check_launch("ground_bwd_ws_kernel")) return rc;
        g_vis = nullptr;
    }
#undef VLG_WS0
#undef VLG_WS
    if (!g_txt && !g_vis) return 0;
    int split = B >= 32 ? 2 : 1;   // two blocks per caption / image (two-addend atomics stay order-free)
    if (const char* e = VLG_ENV("VLG_GD_SPLIT")) split = atoi(e);
    // wide caption side: B blocks x split must cover the chip, and there are B * n_kc step units to share: deeper splits write
    // partial sums (fixed-order reduce), not atomics
    const int n_kc_v = KtV / 128;
    int split_txt = split;
    if (wide) split_txt = ground_dense_split_txt(B, V);
    const bool part_txt = wide && split_txt > 2;
    if (split > 1 || (split_txt > 1 && !part_txt)) {
        hipError_t e = hipSuccess;
        if (g_txt && !part_txt && split_txt > 1) e = hipMemsetAsync(g_txt, 0, sizeof(float) * (size_t)B * Q * kGdD, s);
        if (e == hipSuccess && g_vis && split > 1 && !wide) e = hipMemsetAsync(g_vis, 0, sizeof(float) * (size_t)B * V * kGdD, s);
        if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
#define VLG_GD2(SIDEV, NKCV, MTV, RWV, SEGV, PPV, GRID, KTOT, NKC_RT, STRIDE, FT, OUT)                                  \
    do {                                                                                                               \
        auto kern = ground_bwd_dense_kernel<SIDEV, NKCV, MTV, RWV, SEGV, PPV>;                                         \
        const size_t nb = lds(NKCV * 32, MTV, PPV);                                                                    \
        if (nb > 64 * 1024) {                                                                                          \
            hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nb); \
            if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));            \
        }                                                                                                              \
        hipLaunchKernelGGL(kern, GRID, dim3(256), nb, s, FT, gV, argV, gQ, argQ, coef, B, Q, V, KTOT, NKC_RT, (size_t)(STRIDE), OUT); \
    } while (0)
    // SEGL: 16-byte segments of a feature row that are staged; config-2's 36 regions need 5 of the 8
#define VLG_GD(SIDEV, NKCV, MTV, RWV, FT, OUT)                                                                         \
    do {                                                                                                               \
        if (SIDEV == 0 && V <= 40) VLG_GD2(SIDEV, NKCV, MTV, RWV, (SIDEV == 0 ? 5 : 12), 2, dim3(B, split), NKCV * 32, 1, 0, FT, OUT); \
        else VLG_GD2(SIDEV, NKCV, MTV, RWV, (SIDEV == 0 ? 8 : 12), 2, dim3(B, split), NKCV * 32, 1, 0, FT, OUT);        \
    } while (0)
    if (g_txt) {   // rows = queries (Q <= 96), contraction over regions
        hipLaunchKernelGGL(ground_transpose_kernel, dim3(B, KtV / 32), dim3(256), 0, s, (const uint16_t*)vis, V, KtV, visT);
        if (wide) {
            float* dst = part_txt ? partial : g_txt;
            VLG_GD2(0, 4, 6, 2, 16, 1, dim3(B, split_txt), KtV, n_kc_v, part_txt ? (size_t)B * Q * kGdD : 0, visT, dst);
            if (part_txt) {
                const size_t n4 = (size_t)B * Q * kGdD / 4;
                hipLaunchKernelGGL(ground_partial_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, partial, split_txt, n4,
                                   g_txt);
            }
        } else {
            switch ((Q + 15) / 16) {
                case 1: VLG_GD(0, 2, 1, 1, visT, g_txt); break;
                case 2: VLG_GD(0, 2, 2, 2, visT, g_txt); break;
                case 3: VLG_GD(0, 2, 3, 1, visT, g_txt); break;
                case 4: VLG_GD(0, 2, 4, 2, visT, g_txt); break;
                case 5: VLG_GD(0, 2, 5, 2, visT, g_txt); break;
                default: VLG_GD(0, 2, 6, 2, visT, g_txt); break;
            }
        }
    }
    if (g_vis) {   // rows = regions, contraction over queries (Q <= 96)
        hipLaunchKernelGGL(ground_transpose_kernel, dim3(B, KpQ / 32), dim3(256), 0, s, (const uint16_t*)txt, Q, KpQ, txtT);
        if (wide) {
            VLG_GD2(1, 3, 6, 2, 12, 1, dim3(B, 1, (V + 95) / 96), KpQ, 1, 0, txtT, g_vis);   // one pair per step: two blocks fit a CU's LDS
        } else {
            switch ((V + 15) / 16) {
                case 1: VLG_GD(1, 3, 1, 1, txtT, g_vis); break;
                case 2: VLG_GD(1, 3, 2, 2, txtT, g_vis); break;
                case 3: VLG_GD(1, 3, 3, 1, txtT, g_vis); break;
                default: VLG_GD(1, 3, 4, 2, txtT, g_vis); break;
            }
        }
    }
#undef VLG_GD
#undef VLG_GD2
    return 0;
}

// ---- gather_logit_reduced (joint.py:421-432) on the same machinery ----------------------------------------------------
// Block = caption b; wave = image a (strided); lanes over the queries, then a fixed xor tree: same bits every run.
__global__ __launch_bounds__(256) void reduced_logit_kernel(const float* __restrict__ maxV, const float* __restrict__ marg,
                                                            int B, int Q, float* __restrict__ sums, float* __restrict__ logit) {
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, A = B;
    const float* w = marg + (size_t)b * Q;
    float den = 0.f;
    for (int q = lane; q < Q; q += 64) den += w[q];
#pragma unroll
    for (int k = 1; k < 64; k <<= 1) den += __shfl_xor(den, k, 64);
    if (threadIdx.x == 0) sums[b] = den;
    for (int a = wave; a < A; a += 4) {
        const float* m = maxV + ((size_t)b * A + a) * Q;
        float num = 0.f;
        for (int q = lane; q < Q; q += 64) num += m[q] * w[q];
#pragma unroll
        for (int k = 1; k < 64; k <<= 1) num += __shfl_xor(num, k, 64);
        if (lane == 0) logit[(size_t)b * A + a] = num / den;
    }
}

// gV[b,a,q] = g_logit[b,a] * marginal[b,q] / sum[b] where both masks are on at the arg-max (masked_fill_ passes nothing), else 0.
__global__ __launch_bounds__(256) void reduced_coef_kernel(const float* __restrict__ g_logit, const float* __restrict__ marg,
                                                           const float* __restrict__ sums, const uint16_t* __restrict__ argV,
                                                           const uint8_t* __restrict__ tmask, const uint8_t* __restrict__ vmask,
                                                           int B, int Q, int V, float* __restrict__ gV, float* __restrict__ coef) {
    const int A = B;
    const size_t n = (size_t)B * A * Q;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t pair = i / Q;
        const int q = (int)(i - pair * Q), b = (int)(pair / A), a = (int)(pair - (size_t)b * A);
        const bool open_ = (!tmask || tmask[(size_t)b * Q + q]) && (!vmask || vmask[(size_t)a * V + argV[i]]);
        gV[i] = open_ ? g_logit[pair] * marg[(size_t)b * Q + q] / sums[b] : 0.f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { coef[0] = 1.f; coef[1] = 0.f; }   // max-over-V terms only
}

ReducedPlan::ReducedPlan(int B, int Q) {
    const size_t nV = (size_t)B * B * Q;
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };
    off_maxV = 0;
    off_gV = up(nV);
    off_sum = off_gV + up(nV);
    off_coef = off_sum + up((size_t)B);
    off_argV = off_coef + 64;
    bytes = sizeof(float) * (off_argV + up((nV + 1) / 2));
}

int launch_reduced_logit(const float* maxV, const float* marg, int B, int Q, float* sums, float* logit, hipStream_t s) {
    hipLaunchKernelGGL(reduced_logit_kernel, dim3(B), dim3(256), 0, s, maxV, marg, B, Q, sums, logit);
    return check_launch("reduced_logit_kernel");
}

int launch_reduced_backward(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marg,
                            const float* g_logit, int B, int Q, int V, int d, int in_dtype, float* ws, const ReducedPlan& p,
                            float* g_txt, float* g_vis, hipStream_t s) {
    float *gV = ws + p.off_gV, *coef = ws + p.off_coef;
    const uint16_t* aV = reinterpret_cast<const uint16_t*>(ws + p.off_argV);
    const size_t n = (size_t)B * B * Q;
    hipLaunchKernelGGL(reduced_coef_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65535)), dim3(256), 0, s, g_logit, marg,
                       ws + p.off_sum, aV, tmask, vmask, B, Q, V, gV, coef);
    // the max-over-Q arrays are never read (their factor coef[1] is 0): any valid pointers do
    const int rc = in_dtype == VLG_F32 ? launch_bwd<F32In>(txt, vis, gV, aV, gV, aV, coef, B, Q, V, d, g_txt, g_vis, s, false)
                                       : launch_bwd<BF16In>(txt, vis, gV, aV, gV, aV, coef, B, Q, V, d, g_txt, g_vis, s, false);
    if (rc) return rc;
    return check_launch("align_reduced_backward");
}

GroundPlan::GroundPlan(int B, int Q, int V) {
    const size_t nV = (size_t)B * B * Q, nQ = (size_t)B * B * V;
    auto up = [](size_t x) { return (x + 63) & ~(size_t)63; };   // 256-byte aligned float offsets
    off_maxV = 0;
    off_maxQ = up(nV);
    off_part = off_maxQ + up(nQ);
    off_coef = off_part + up(2 * (size_t)B * kCeMaxY);
    off_argV = off_coef + 64;                          // uint16 arrays, offsets still counted in floats
    off_argQ = off_argV + up((nV + 1) / 2);
    // dense backward (bf16, Q <= 96): transposed bf16 features [B][128][KtV] + [B][128][96], and for the wide layouts (V > 64)
    // the caption side's partial sums [split][B][Q][128] fp32
    off_featT = off_argQ + up((nQ + 1) / 2);
    const bool dense = Q <= 96;
    const size_t KtV = V > 64 ? (size_t)(V + 127) / 128 * 128 : 64;
    off_partial = off_featT + (dense ? up((size_t)B * 128 * (KtV + 96) / 2) : 0);
    const size_t partial = dense && V > 64 ? (size_t)ground_dense_split_txt(B, V) * B * Q * 128 : 0;
    // float32 features on two fp16 parts (align_argmax_kernel<NP = 2>): hi | lo of txt and vis (d <= 128), 2 x 64 partial maxima, 2 inverse scales
    off_parts = off_partial + up(partial);
    bytes = sizeof(float) * (off_parts + up((size_t)B * (Q + V) * 128) + 256);
}

int launch_grounding_tail(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marg,
                          int B, int Q, int V, int d, int in_dtype, float num_token, float w_v2t, float* ws,
                          const GroundPlan& p, float* out_sums, float* g_txt, float* g_vis, hipStream_t s) {
    float *mV = ws + p.off_maxV, *mQ = ws + p.off_maxQ, *part = ws + p.off_part, *coef = ws + p.off_coef;
    const uint16_t* aV = reinterpret_cast<const uint16_t*>(ws + p.off_argV);
    const uint16_t* aQ = reinterpret_cast<const uint16_t*>(ws + p.off_argQ);
    // strips of columns per caption / image: [B x cw] floats staged in LDS, at most kCeMaxY strips
    struct Strips { int cw, y; size_t lds; };
    auto strips = [&](int ncols) {
        const int budget = std::max(16, (int)(48 * 1024 / (4 * (size_t)B)));
        int y = std::min(kCeMaxY, (ncols + budget - 1) / budget);
        if (B < 256) y = std::max(y, std::min(std::min(kCeMaxY, (ncols + 31) / 32), (256 + B - 1) / B));   // few rows: still cover the chip
        const int cw = (ncols + y - 1) / y;
        y = (ncols + cw - 1) / cw;
        return Strips{cw, y, sizeof(float) * ((size_t)B * (cw | 1) + kCeTileThreads + 3 * (size_t)cw)};
    };
    const Strips s1 = strips(Q), s2 = strips(V);
    const bool tiled = s1.lds <= 144 * 1024 && s2.lds <= 144 * 1024 && s1.cw <= kCeTileThreads && s2.cw <= kCeTileThreads && !VLG_ENV("VLG_GROUND_CE_OLD");
    // column shares for the streaming kernel: only when B alone leaves CUs idle and there are several 256-column strips
    auto shares = [&](int ncols) {
        const int st = (ncols + kCeThreads - 1) / kCeThreads;
        return B >= 256 ? 1 : std::max(1, std::min(std::min(st, kCeMaxY), (256 + B - 1) / B));
    };
    const int y1 = tiled ? s1.y : shares(Q), y2 = tiled ? s2.y : shares(V);
    float* part2 = part + (size_t)B * y1;
    // txt2vis: fixed = caption b, rows = images a: x[a][q] = mV[(b*A + a)*Q + q]; gate: tmask[b][q], vmask[a][argV]
    // vis2txt: fixed = image a, rows = captions b: x[b][v] = mQ[(b*A + a)*V + v]; weights = vis_mask as 0/1 (joint.py:481,
    // null = all ones); gate: vmask[a][v], tmask[b][argQ]
    if (tiled) {
        auto k1 = ground_ce_tile_kernel<float>;
        auto k2 = ground_ce_tile_kernel<uint8_t>;
        for (auto kp : {std::make_pair((const void*)k1, s1.lds), std::make_pair((const void*)k2, s2.lds)})
            if (kp.second > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute(kp.first, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kp.second);
                if (e != hipSuccess) return set_error((int)e, "grounding_loss: hipFuncSetAttribute(%zu): %s", kp.second, hipGetErrorString(e));
            }
        if (!VLG_ENV("VLG_GROUND_CE_SPLIT")) {
            const size_t lds2 = std::max(s1.lds, s2.lds);
            if (lds2 > 64 * 1024) {
                hipError_t e = hipFuncSetAttribute((const void*)ground_ce_tile2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
                if (e != hipSuccess) return set_error((int)e, "grounding_loss: hipFuncSetAttribute(%zu): %s", lds2, hipGetErrorString(e));
            }
            const CeSide sa{mV, (size_t)B * Q, (size_t)Q, Q, s1.cw, marg, aV, tmask, vmask, V, part, y1};
            const CeSide sb{mQ, (size_t)V, (size_t)B * V, V, s2.cw, vmask, aQ, vmask, tmask, Q, part2, y2};
            hipLaunchKernelGGL(ground_ce_tile2_kernel, dim3(B, y1 + y2), dim3(kCeTileThreads), lds2, s, B, sa, sb);
        } else {
        hipLaunchKernelGGL(k1, dim3(B, y1), dim3(kCeTileThreads), s1.lds, s, mV, (size_t)B * Q, (size_t)Q, B, Q, s1.cw, marg, aV, tmask, vmask,
                           V, part);
        hipLaunchKernelGGL(k2, dim3(B, y2), dim3(kCeTileThreads), s2.lds, s, mQ, (size_t)V, (size_t)B * V, B, V, s2.cw, vmask, aQ, vmask, tmask,
                           Q, part2);
        }
    } else {
        hipLaunchKernelGGL(ground_ce_kernel<float>, dim3(B, y1), dim3(kCeThreads), 0, s, mV, (size_t)B * Q, (size_t)Q, B, Q, marg, aV,
                           tmask, vmask, V, part);
        hipLaunchKernelGGL(ground_ce_kernel<uint8_t>, dim3(B, y2), dim3(kCeThreads), 0, s, mQ, (size_t)V, (size_t)B * V, B, V, vmask, aQ,
                           vmask, tmask, Q, part2);
    }
    hipLaunchKernelGGL(ground_sum_kernel, dim3(1), dim3(64), 0, s, part, part2, B * y1, B * y2, num_token, w_v2t, out_sums, coef);
    if ((g_txt || g_vis) && in_dtype == VLG_BF16 && d == kGdD && Q <= 96 && V <= 65535 && !VLG_ENV("VLG_GROUND_SPARSE")) {
        // bf16 features, d = 128, up to 96 queries: the dense route on the matrix cores
        if (int rc = launch_bwd_dense(txt, vis, mV, aV, mQ, aQ, coef, B, Q, V, reinterpret_cast<uint16_t*>(ws + p.off_featT),
                                      ws + p.off_partial, g_txt, g_vis, s))
            return rc;
    } else if (g_txt || g_vis) {
        const int rc = in_dtype == VLG_F32 ? launch_bwd<F32In>(txt, vis, mV, aV, mQ, aQ, coef, B, Q, V, d, g_txt, g_vis, s, w_v2t > 0.f)
                                           : launch_bwd<BF16In>(txt, vis, mV, aV, mQ, aQ, coef, B, Q, V, d, g_txt, g_vis, s, w_v2t > 0.f);
        if (rc) return rc;
    }
    return check_launch("grounding_loss");
}

}  // namespace vlg
