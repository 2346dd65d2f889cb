// vlg_scorer.hip -- the score half of DiscriminativeNDMV._forward (src/model/ldndmv.py:179-209):
// from the factorised-bilinear scorers' projected inputs straight to the root-merged potentials the DP reads.
//
//   reference                                                                   here
//   attach_scorer(h_parent, h_child)  = einsum('bhdve,bcdve->bhcdv', x1, x2)     x1 [B,L,2,2,r] = project1(h_parent),
//     (nn/dmv_spec.py:66-76)          -> [B,L,T,2,2]                              x2 [T,2,2,r]   = project2(h_child)
//   .log_softmax(2)                   over the T tokens            (:185)        lse[b,h,d,v] over t, never stored per token
//   .gather(2, token)                 -> [B,L,L,2,2]               (:189-190)    dot(x1[b,h,dir,v], x2[token[b,c],dir,v]) - lse
//   tril / triu direction select      -> [B,L,L,2]                 (:191-194)    dir = LEFT if c < h, RIGHT if c > h, 0 on the diagonal
//   masked_fill_(function heads)                                   (:195-199)    head_mask[b,h] -> the row is `mask_fill`
//   dec_scorer(h_parent, h_dec).permute.log_softmax(-1)            (:201)        y1 [B,L,2,2,r], y2 [2,2,2,r]: 2-way log-softmax
//   gather(root_prob, 1, token)                                    (:205-207)    root_rule[token[b,c]]  (root_rule [T] is batch-free)
//   DMV1o.merge(dec, attach, root)                                 (:209)        written in merged layout [B,N,2,2,2] / [B,N,N,2]
// The [B,L,T,2,2] rule table (7.4 MB at B=256, L=40, T=45), its gathered / masked / merged copies and the ~10 launches
// between them do not exist; neither does their autograd tape: the adjoint kernel goes from the cotangents of the merged
// potentials (the DP's expected counts) back to x1, x2, y1, y2, root_rule, recomputing the softmax weights.
//
// The rank r is small (16 in config/model/vlgae.yaml:10,115-117): every product here is a length-r dot -- fp32 VALU work out of
// LDS, not a matrix-core shape.  One workgroup per sentence; parameter gradients leave as per-sentence partials that a second
// launch adds in sentence order (no atomics, bit-reproducible).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_dp_core.h"

namespace vlg {

namespace {

constexpr int kScThreads = 1024;   // 16 wavefronts per sentence: the loops are LDS-latency chains, more waves hide them
constexpr float kMergeZero = -1e12f;   // DMV1o.merge's `zero` (distributions.py:253, bound at import: semirings.py:16)

constexpr int kScLdsFloats = 160 * 1024 / 4;
struct ScLayout {   // LDS carving, in floats
    int L, T, r, rp;
    int x1, x2, y1, y2, lse, tok, hm, tot, coef, total;
    int HC;   // backward: head positions per pass of the coefficient table (L when everything fits; fewer -> several passes)
    __host__ __device__ ScLayout(int L_, int T_, int r_, bool bwd) : L(L_), T(T_), r(r_) {
        rp = r | 1;                        // odd row pitch: rows of different tokens on different banks
        int o = 0;
        x1 = o; o += L * 4 * rp;
        x2 = o; o += T * 4 * rp;
        y1 = o; o += L * 4 * rp;
        y2 = o; o += 2 * 4 * rp;
        lse = o; o += L * 4;
        tok = o; o += L;
        hm = o; o += L;
        tot = o; o += bwd ? L * 4 * 3 : 0;   // backward: tot[h][dv], ds_dec[h][dv][2]
        coef = o;                            // backward: cotangent of score[h][dv][t] for HC head positions at a time
        HC = L;
        if (bwd) {
            const int room = (kScLdsFloats - o) / (4 * T);
            HC = room >= L ? L : (room > 0 ? room : 0);
            o += HC * 4 * T;
        }
        total = HC > 0 ? o : kScLdsFloats + 1;   // (not even one head position: refused by check_shape)
    }
};

// rows of r values, `ld` elements apart (ld = r: contiguous; wider: column slices of a GEMM output holding several projections)
struct ScLd { int x1, x2, y1, y2; };

template <typename In>
__device__ __forceinline__ void load_rows(const typename In::T* src, size_t n_rows, int r, int rp, float* dst, int tid, int ld) {
    for (int i = tid; i < (int)n_rows * r; i += kScThreads) {
        const int row = i / r, e = i - row * r;
        dst[row * rp + e] = In::ld(src, (size_t)row * ld + e);
    }
}

__device__ __forceinline__ void st_grad(float* p, size_t i, float v) { p[i] = v; }
__device__ __forceinline__ void st_grad(uint16_t* p, size_t i, float v) {   // bf16, round to nearest even
    const uint32_t u = __float_as_uint(v);
    p[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// length-r dot of two LDS rows; R > 0: compile-time rank (all 2R reads issued together), R = 0: run-time loop
template <int R>
__device__ __forceinline__ float dot_r(const float* a, const float* b, int r) {
    if constexpr (R > 0) {
        float av[R], bv[R];
#pragma unroll
        for (int e = 0; e < R; ++e) { av[e] = a[e]; bv[e] = b[e]; }
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int e = 0; e < R; e += 2) { s0 = fmaf(av[e], bv[e], s0); s1 = fmaf(av[e + 1], bv[e + 1], s1); }
        return s0 + s1;
    } else {
        float s = 0.f;
        for (int e = 0; e < r; ++e) s = fmaf(a[e], b[e], s);
        return s;
    }
}

// lse[h][dv] = logsumexp_t dot(x1[h][dv], x2[t][dv]): 4 lanes per (h, dv) pair split the tokens, then combine
template <int R>
__device__ __forceinline__ void compute_lse(const float* x1s, const float* x2s, float* lse, int L, int T, int r, int rp, int tid) {
    for (int p0 = 0; p0 < L * 4; p0 += kScThreads / 4) {
        const int p = p0 + (tid >> 2), part = tid & 3;
        const bool live = p < L * 4;
        const int pc = live ? p : 0, dv = pc & 3;
        const float* a = x1s + pc * rp;
        float m = -3.0e38f, s = 0.f;
        for (int t = part; t < T; t += 4) {   // online log-sum-exp
            const float v = dot_r<R>(a, x2s + (t * 4 + dv) * rp, r);
            const float nm = fmaxf(m, v);
            s = s * __expf(m - nm) + __expf(v - nm);
            m = nm;
        }
#pragma unroll
        for (int k = 1; k <= 2; k <<= 1) {
            const float om = __shfl_xor(m, k, 64), os = __shfl_xor(s, k, 64);
            const float nm = fmaxf(m, om);
            s = s * __expf(m - nm) + os * __expf(om - nm);
            m = nm;
        }
        if (live && part == 0) lse[p] = m + __logf(s);
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
template <typename In, typename Out, int R>
__global__ __launch_bounds__(kScThreads) void scorer_fwd_kernel(
    const typename In::T* __restrict__ x1, const typename In::T* __restrict__ x2, const typename In::T* __restrict__ y1,
    const typename In::T* __restrict__ y2, const float* __restrict__ root_rule, const int64_t* __restrict__ token,
    const uint8_t* __restrict__ head_mask, int L, int T, int r, ScLd ld, float mask_fill, Out* __restrict__ mdec, Out* __restrict__ matt) {
    extern __shared__ float smem[];
    const ScLayout lay(L, T, r, false);
    const int b = blockIdx.x, tid = threadIdx.x, N = L + 1, rp = lay.rp;
    float *x1s = smem + lay.x1, *x2s = smem + lay.x2, *y1s = smem + lay.y1, *y2s = smem + lay.y2, *lse = smem + lay.lse;
    int* tok = reinterpret_cast<int*>(smem + lay.tok);
    int* hm = reinterpret_cast<int*>(smem + lay.hm);
    load_rows<In>(x1 + (size_t)b * L * 4 * ld.x1, (size_t)L * 4, r, rp, x1s, tid, ld.x1);
    load_rows<In>(x2, (size_t)T * 4, r, rp, x2s, tid, ld.x2);
    load_rows<In>(y1 + (size_t)b * L * 4 * ld.y1, (size_t)L * 4, r, rp, y1s, tid, ld.y1);
    load_rows<In>(y2, 8, r, rp, y2s, tid, ld.y2);
    for (int i = tid; i < L; i += kScThreads) {
        tok[i] = (int)min(max(token[(size_t)b * L + i], (int64_t)0), (int64_t)(T - 1));
        hm[i] = head_mask ? head_mask[(size_t)b * L + i] : 0;
    }
    __syncthreads();
    compute_lse<R>(x1s, x2s, lse, L, T, r, rp, tid);
    __syncthreads();
    // ---- merged attach [N][N][2]: (h, c) positions in the root-augmented sentence ----
    Out* ma = matt + (size_t)b * N * N * 2;
    for (int i = tid; i < N * N; i += kScThreads) {
        const int hn = i / N, cn = i - hn * N;
        float v0 = kMergeZero, v1 = kMergeZero;                       // HASCHILD, NOCHILD
        if (cn >= 1) {
            const int c = cn - 1;
            if (hn == 0) v1 = root_rule[tok[c]];                       // attach[:, 0, 1:, NOCHILD] = root (distributions.py:262)
            else {
                const int h = hn - 1;
                if (hm[h]) v0 = v1 = mask_fill;                        // function-word head: the whole row (ldndmv.py:195-199)
                else if (c == h) v0 = v1 = 0.f;                        // both masks are zero on the diagonal (:191-194)
                else {
                    const int d = c < h ? 0 : 1;                       // LEFT = 0, RIGHT = 1
                    const float* xa = x1s + (h * 4 + d * 2) * rp;
                    const float* xb = x2s + (tok[c] * 4 + d * 2) * rp;
                    v0 = dot_r<R>(xa, xb, r) - lse[h * 4 + d * 2];
                    v1 = dot_r<R>(xa + rp, xb + rp, r) - lse[h * 4 + d * 2 + 1];
                }
            }
        }
        ma[(size_t)i * 2] = (Out)v0;
        ma[(size_t)i * 2 + 1] = (Out)v1;
    }
    // ---- merged dec [N][2][2][2]: row 0 = zero except [RIGHT] = one (distributions.py:259-260); rows 1.. = 2-way log-softmax ----
    Out* md = mdec + (size_t)b * N * 8;
    for (int i = tid; i < N * 4; i += kScThreads) {
        const int hn = i >> 2, dv = i & 3;
        float g, s;
        if (hn == 0) g = s = (dv >> 1) == 1 ? 0.f : kMergeZero;
        else {
            const float* ya = y1s + ((hn - 1) * 4 + dv) * rp;
            const float a0 = dot_r<R>(ya, y2s + (0 * 4 + dv) * rp, r), a1 = dot_r<R>(ya, y2s + (1 * 4 + dv) * rp, r);
            const float m = fmaxf(a0, a1), z = m + __logf(__expf(a0 - m) + __expf(a1 - m));
            g = a0 - z;
            s = a1 - z;
        }
        md[(size_t)i * 2] = (Out)g;
        md[(size_t)i * 2 + 1] = (Out)s;
    }
}

// --------------------------------------------------------------------------------------------------------------- backward
// g_matt [B,N,N,2], g_mdec [B,N,2,2,2] fp32 -> d_x1, d_y1 [B,L,2,2,r] (G: fp32 or bf16, rows ld_dx1 / ld_dy1 apart) and per-sentence
// partials of d_x2 [T,2,2,r], d_y2 [2,2,2,r], d_root_rule [T] (part [B][T*4*r + 8*r + T]).
template <typename In, int R, typename G>
__global__ __launch_bounds__(kScThreads) void scorer_bwd_kernel(
    const typename In::T* __restrict__ x1, const typename In::T* __restrict__ x2, const typename In::T* __restrict__ y1,
    const typename In::T* __restrict__ y2, const int64_t* __restrict__ token, const uint8_t* __restrict__ head_mask,
    const float* __restrict__ g_mdec, const float* __restrict__ g_matt, int L, int T, int r, ScLd ld, G* __restrict__ d_x1, int ld_dx1,
    G* __restrict__ d_y1, int ld_dy1, float* __restrict__ part) {
    extern __shared__ float smem[];
    const ScLayout lay(L, T, r, true);
    const int b = blockIdx.x, tid = threadIdx.x, N = L + 1, rp = lay.rp;
    float *x1s = smem + lay.x1, *x2s = smem + lay.x2, *y1s = smem + lay.y1, *y2s = smem + lay.y2, *lse = smem + lay.lse;
    float* tot = smem + lay.tot;            // [L*4]
    float* dsd = tot + L * 4;               // [L*4][2]: cotangent of the dec scores
    int* tok = reinterpret_cast<int*>(smem + lay.tok);
    int* hm = reinterpret_cast<int*>(smem + lay.hm);
    load_rows<In>(x1 + (size_t)b * L * 4 * ld.x1, (size_t)L * 4, r, rp, x1s, tid, ld.x1);
    load_rows<In>(x2, (size_t)T * 4, r, rp, x2s, tid, ld.x2);
    load_rows<In>(y1 + (size_t)b * L * 4 * ld.y1, (size_t)L * 4, r, rp, y1s, tid, ld.y1);
    load_rows<In>(y2, 8, r, rp, y2s, tid, ld.y2);
    for (int i = tid; i < L; i += kScThreads) {
        tok[i] = (int)min(max(token[(size_t)b * L + i], (int64_t)0), (int64_t)(T - 1));
        hm[i] = head_mask ? head_mask[(size_t)b * L + i] : 0;
    }
    __syncthreads();
    compute_lse<R>(x1s, x2s, lse, L, T, r, rp, tid);
    const float* ga = g_matt + (size_t)b * N * N * 2;
    const float* gd = g_mdec + (size_t)b * N * 8;
    // cotangent of attach[h][c][v] (0 for masked heads and on the diagonal): cnt(h, c, v)
    auto cnt = [&](int h, int c, int v) -> float { return (hm[h] || c == h) ? 0.f : ga[((size_t)(h + 1) * N + (c + 1)) * 2 + v]; };
    float* coef = smem + lay.coef;          // [L*4][T]: d loss / d score[h][dv][t] = [children with token t] - p_t * tot
    // tot[h][dv] = sum over the children on side d;  dsd[h][dv][k] = g[k] - p[k] (g0 + g1)
    for (int p = tid; p < L * 4; p += kScThreads) {
        const int h = p >> 2, d = (p >> 1) & 1, v = p & 1;
        float s = 0.f;
        for (int c = d ? h + 1 : 0; c < (d ? L : h); ++c) s += cnt(h, c, v);
        tot[p] = s;
        const float* ya = y1s + p * rp;
        const float a0 = dot_r<R>(ya, y2s + (0 * 4 + (p & 3)) * rp, r), a1 = dot_r<R>(ya, y2s + (1 * 4 + (p & 3)) * rp, r);
        const float m = fmaxf(a0, a1), e0 = __expf(a0 - m), e1 = __expf(a1 - m), inv = 1.f / (e0 + e1);
        const float g0 = gd[(size_t)(h + 1) * 8 + (p & 3) * 2], g1 = gd[(size_t)(h + 1) * 8 + (p & 3) * 2 + 1];
        dsd[p * 2] = g0 - e0 * inv * (g0 + g1);
        dsd[p * 2 + 1] = g1 - e1 * inv * (g0 + g1);
    }
    __syncthreads();
    float* pt = part + (size_t)b * ((size_t)T * 4 * r + 8 * r + T);
    // The coefficient table for the head positions [hb, he) at a time: one pass when it fits beside the rows (HC = L: every configuration
    // of BASELINE.json), else several -- d_x2's partial then adds the passes in order (its thread owns the element in every pass).
    const int HC = lay.HC;
    for (int hb = 0; hb < L; hb += HC) {
        const int he = min(L, hb + HC), p0 = hb * 4, np = (he - hb) * 4;
        for (int i = tid; i < np * T; i += kScThreads) {      // - softmax weight * tot
            const int pl = i / T, t = i - pl * T, p = p0 + pl;
            const float tt = tot[p];
            coef[i] = tt != 0.f ? -tt * __expf(dot_r<R>(x1s + p * rp, x2s + (t * 4 + (p & 3)) * rp, r) - lse[p]) : 0.f;
        }
        __syncthreads();
        for (int pl = tid; pl < np; pl += kScThreads) {       // + the children's counts at their tokens; row p has one owner, c ascending
            const int p = p0 + pl, h = p >> 2, d = (p >> 1) & 1, v = p & 1;
            for (int c = d ? h + 1 : 0; c < (d ? L : h); ++c) coef[pl * T + tok[c]] += cnt(h, c, v);
        }
        __syncthreads();
        // ---- d_x1[h][dv][:] = sum_t coef[t] x2[t][dv][:];  d_y1[h][dv][:] = sum_k dsd[k] y2[k][dv][:] ----
        for (int i = tid; i < np * r; i += kScThreads) {
            const int pl = i / r, e = i - pl * r, p = p0 + pl, dv = p & 3;
            float acc = 0.f;
#pragma unroll 8
            for (int t = 0; t < T; ++t) acc = fmaf(coef[pl * T + t], x2s[(t * 4 + dv) * rp + e], acc);
            st_grad(d_x1, ((size_t)b * L * 4 + p) * ld_dx1 + e, acc);
            st_grad(d_y1, ((size_t)b * L * 4 + p) * ld_dy1 + e, dsd[p * 2] * y2s[(0 * 4 + dv) * rp + e] + dsd[p * 2 + 1] * y2s[(1 * 4 + dv) * rp + e]);
        }
        // ---- partials of the batch-shared tables ----
        for (int i = tid; i < T * 4 * r; i += kScThreads) {   // d_x2[t][dv][e] = sum_h coef[h][dv][t] x1[h][dv][e], h ascending
            const int q = i / r, e = i - q * r, t = q >> 2, dv = q & 3;
            float acc = 0.f;
#pragma unroll 8
            for (int h = hb; h < he; ++h) acc = fmaf(coef[((h - hb) * 4 + dv) * T + t], x1s[(h * 4 + dv) * rp + e], acc);
            pt[i] = hb == 0 ? acc : pt[i] + acc;
        }
        if (he < L) __syncthreads();                          // (the next pass overwrites the table)
    }
    float* pty = pt + (size_t)T * 4 * r;
    for (int i = tid; i < 8 * r; i += kScThreads) {   // d_y2[k][dv][e] = sum_h dsd[h][dv][k] y1[h][dv][e]
        const int q = i / r, e = i - q * r, k = q >> 2, dv = q & 3;
        float acc = 0.f;
        for (int h = 0; h < L; ++h) acc = fmaf(dsd[(h * 4 + dv) * 2 + k], y1s[(h * 4 + dv) * rp + e], acc);
        pty[i] = acc;
    }
    float* ptr_ = pty + 8 * r;
    for (int t = tid; t < T; t += kScThreads) {       // d_root_rule[t] = sum_{c: tok c = t} g_matt[0][c+1][NOCHILD]
        float acc = 0.f;
        for (int c = 0; c < L; ++c)
            if (tok[c] == t) acc += ga[((size_t)(c + 1)) * 2 + 1];
        ptr_[t] = acc;
    }
}

// out[i] = sum_b part[b][i] in a fixed order: 64 outputs x 4 sentence groups per block (group g adds b = g, g + 4, ...
// ascending), then (g0 + g1) + (g2 + g3)
template <typename G>
__global__ __launch_bounds__(256) void scorer_reduce_kernel(const float* __restrict__ part, int B, int n, G* __restrict__ d_x2,
                                                            int n_x2, G* __restrict__ d_y2, int n_y2, float* __restrict__ d_root) {
    __shared__ float sm[4][64];
    const int i = blockIdx.x * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float t = 0.f;
    if (i < n) {
        int b = grp;
        for (; b + 28 < B; b += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 4 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) t += v[u];
        }
        for (; b < B; b += 4) t += part[(size_t)b * n + i];
    }
    sm[grp][threadIdx.x & 63] = t;
    __syncthreads();
    if (grp != 0 || i >= n) return;
    t = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
    if (i < n_x2) st_grad(d_x2, (size_t)i, t);
    else if (i < n_x2 + n_y2) st_grad(d_y2, (size_t)(i - n_x2), t);
    else d_root[i - n_x2 - n_y2] = t;
}

int check_shape(const char* what, int B, int L, int T, int r, bool bwd, size_t* lds) {
    if (B < 0 || L < 1 || T < 1 || r < 1) return set_error(VLG_ERR_SHAPE, "%s: bad shape B=%d L=%d T=%d r=%d", what, B, L, T, r);
    *lds = sizeof(float) * (size_t)ScLayout(L, T, r, bwd).total;
    if (*lds > 160 * 1024)
        return set_error(VLG_ERR_SHAPE, "%s: L=%d T=%d r=%d need %zu bytes of LDS (limit 160 KiB): the sentence's and the tokens' projected rows do not fit one workgroup",
                         what, L, T, r, *lds);
    return 0;
}

template <typename K>
int prep_lds(K kernel, size_t lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    }
    return 0;
}

}  // namespace

}  // namespace vlg

extern "C" {

int vlg_ndmv_potentials(const void* x1, int ld_x1, const void* x2, int ld_x2, const void* y1, int ld_y1, const void* y2, int ld_y2,
                        const float* root_rule, const int64_t* token, const uint8_t* head_mask, int B, int L, int T, int r, int in_dtype,
                        float mask_fill, int out_dtype, void* merged_dec, void* merged_attach, void* stream) {
    using namespace vlg;
    size_t lds;
    if (int rc = check_shape("ndmv_potentials", B, L, T, r, false, &lds)) return rc;
    if (ld_x1 < r || ld_x2 < r || ld_y1 < r || ld_y2 < r)
        return set_error(VLG_ERR_SHAPE, "ndmv_potentials: row strides %d %d %d %d below r=%d", ld_x1, ld_x2, ld_y1, ld_y2, r);
    const ScLd ld{ld_x1, ld_x2, ld_y1, ld_y2};
    if ((in_dtype != VLG_F32 && in_dtype != VLG_BF16) || (out_dtype != VLG_F32 && out_dtype != VLG_BF16))
        return set_error(VLG_ERR_DTYPE, "ndmv_potentials: dtypes %d -> %d", in_dtype, out_dtype);
    if (B == 0) return 0;
    if (!x1 || !x2 || !y1 || !y2 || !root_rule || !token || !merged_dec || !merged_attach) return set_error(VLG_ERR_ARG, "ndmv_potentials: null buffer");
    hipStream_t s = (hipStream_t)stream;
#define VLG_GO_R(IN, OUT, RR)                                                                                                       \
    {                                                                                                                               \
        auto k = scorer_fwd_kernel<IN, OUT, RR>;                                                                                    \
        if (int rc = prep_lds(k, lds)) return rc;                                                                                   \
        hipLaunchKernelGGL(k, dim3(B), dim3(kScThreads), lds, s, (const IN::T*)x1, (const IN::T*)x2, (const IN::T*)y1, (const IN::T*)y2, \
                           root_rule, token, head_mask, L, T, r, ld, mask_fill, (OUT*)merged_dec, (OUT*)merged_attach);            \
    }
#define VLG_GO(IN, OUT)                                  \
    {                                                    \
        if (r == 16) VLG_GO_R(IN, OUT, 16)               \
        else if (r == 8) VLG_GO_R(IN, OUT, 8)            \
        else if (r == 32) VLG_GO_R(IN, OUT, 32)          \
        else VLG_GO_R(IN, OUT, 0)                        \
    }
    if (in_dtype == VLG_F32 && out_dtype == VLG_F32) VLG_GO(F32In, float)
    else if (in_dtype == VLG_F32) VLG_GO(F32In, __bf16)
    else if (out_dtype == VLG_F32) VLG_GO(BF16In, float)
    else VLG_GO(BF16In, __bf16)
#undef VLG_GO
#undef VLG_GO_R
    return check_launch("scorer_fwd_kernel");
}

size_t vlg_ndmv_potentials_backward_workspace(int B, int L, int T, int r) {
    if (B < 1 || L < 1 || T < 1 || r < 1) return 0;
    return sizeof(float) * (size_t)B * ((size_t)T * 4 * r + 8 * (size_t)r + T);
}

int vlg_ndmv_potentials_backward(const void* x1, int ld_x1, const void* x2, int ld_x2, const void* y1, int ld_y1, const void* y2, int ld_y2,
                                 const int64_t* token, const uint8_t* head_mask, const float* g_merged_dec, const float* g_merged_attach,
                                 int B, int L, int T, int r, int in_dtype, void* ws, size_t ws_bytes, int grad_dtype, void* d_x1, int ld_dx1,
                                 void* d_x2, void* d_y1, int ld_dy1, void* d_y2, float* d_root_rule, void* stream) {
    using namespace vlg;
    size_t lds;
    if (int rc = check_shape("ndmv_potentials_backward", B, L, T, r, true, &lds)) return rc;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "ndmv_potentials_backward: in_dtype %d", in_dtype);
    if (grad_dtype != VLG_F32 && grad_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "ndmv_potentials_backward: grad_dtype %d", grad_dtype);
    if (ld_x1 < r || ld_x2 < r || ld_y1 < r || ld_y2 < r || ld_dx1 < r || ld_dy1 < r)
        return set_error(VLG_ERR_SHAPE, "ndmv_potentials_backward: row strides %d %d %d %d / %d %d below r=%d", ld_x1, ld_x2, ld_y1, ld_y2, ld_dx1,
                         ld_dy1, r);
    if (!d_x2 || !d_y2 || !d_root_rule || (B > 0 && (!d_x1 || !d_y1))) return set_error(VLG_ERR_ARG, "ndmv_potentials_backward: null output");
    hipStream_t s = (hipStream_t)stream;
    const int n_x2 = T * 4 * r, n_y2 = 8 * r, n = n_x2 + n_y2 + T;
    const size_t gsz = grad_dtype == VLG_F32 ? sizeof(float) : sizeof(uint16_t);
    if (B == 0) {
        hipError_t e = hipMemsetAsync(d_x2, 0, gsz * n_x2, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_y2, 0, gsz * n_y2, s);
        if (e == hipSuccess) e = hipMemsetAsync(d_root_rule, 0, sizeof(float) * T, s);
        return e == hipSuccess ? 0 : set_error((int)e, "ndmv_potentials_backward: %s", hipGetErrorString(e));
    }
    if (!x1 || !x2 || !y1 || !y2 || !token || !g_merged_dec || !g_merged_attach) return set_error(VLG_ERR_ARG, "ndmv_potentials_backward: null buffer");
    const size_t need = vlg_ndmv_potentials_backward_workspace(B, L, T, r);
    if (!ws || ws_bytes < need) return set_error(VLG_ERR_WORKSPACE, "ndmv_potentials_backward: workspace %zu bytes < %zu", ws_bytes, need);
    const ScLd ld{ld_x1, ld_x2, ld_y1, ld_y2};
#define VLG_GO_R(IN, RR, G)                                                                                                         \
    {                                                                                                                               \
        auto k = scorer_bwd_kernel<IN, RR, G>;                                                                                      \
        if (int rc = prep_lds(k, lds)) return rc;                                                                                   \
        hipLaunchKernelGGL(k, dim3(B), dim3(kScThreads), lds, s, (const IN::T*)x1, (const IN::T*)x2, (const IN::T*)y1, (const IN::T*)y2, \
                           token, head_mask, g_merged_dec, g_merged_attach, L, T, r, ld, (G*)d_x1, ld_dx1, (G*)d_y1, ld_dy1, (float*)ws); \
    }
#define VLG_GO(IN, G)                                  \
    {                                                  \
        if (r == 16) VLG_GO_R(IN, 16, G)               \
        else if (r == 8) VLG_GO_R(IN, 8, G)            \
        else if (r == 32) VLG_GO_R(IN, 32, G)          \
        else VLG_GO_R(IN, 0, G)                        \
    }
    if (in_dtype == VLG_F32 && grad_dtype == VLG_F32) VLG_GO(F32In, float)
    else if (in_dtype == VLG_F32) VLG_GO(F32In, uint16_t)
    else if (grad_dtype == VLG_F32) VLG_GO(BF16In, float)
    else VLG_GO(BF16In, uint16_t)
#undef VLG_GO
#undef VLG_GO_R
    if (int rc = check_launch("scorer_bwd_kernel")) return rc;
    if (grad_dtype == VLG_F32)
        hipLaunchKernelGGL(scorer_reduce_kernel<float>, dim3((n + 63) / 64), dim3(256), 0, s, (const float*)ws, B, n, (float*)d_x2, n_x2, (float*)d_y2,
                           n_y2, d_root_rule);
    else
        hipLaunchKernelGGL(scorer_reduce_kernel<uint16_t>, dim3((n + 63) / 64), dim3(256), 0, s, (const float*)ws, B, n, (uint16_t*)d_x2, n_x2,
                           (uint16_t*)d_y2, n_y2, d_root_rule);
    return check_launch("scorer_reduce_kernel");
}

}  // extern "C"
