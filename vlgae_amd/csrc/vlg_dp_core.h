// vlg_dp_core.h -- per-thread phase bodies of the structured-DP kernels (DMV1o and DepTree).
//
// Each function is the work of ONE thread (tid of nt) in ONE barrier-delimited phase of the
// span-width loop.  The HIP kernels (vlg_dp.hip) call them between __syncthreads(); the
// host-side phase emulator used by the CPU test-suite (tests/emu/emu_dp.cpp) calls the very
// same bodies with host threads standing in for lanes (in several serialisation orders, to expose
// intra-phase races).  Nothing here is a CPU fallback: the product only ever runs these through
// the gfx950 kernels.
//
// Algorithm (reference: /root/reference/src/model/torch_struct/dmv.py:19-66 and
// deptree.py:25-76; their outside pass is autograd).  Notation, h = head:
//   CL(h,l) complete span, head h reaching LEFT to l (l <= h)    stored at C[h*P + l]
//   CR(h,r) complete span, head h reaching RIGHT to r (r >= h)   stored at C[h*P + r + 1]
//   IL(h,c) incomplete span with arc h -> c, c < h               stored at I[h*P + c]
//   IR(h,c) incomplete span with arc h -> c, c > h               stored at I[h*P + c + 1]
// (the "+1 column shift for right-facing items" is the reference's chart layout, dmv.py:32-35).
// DMV cells are float2: .x = valence HASCHILD(0), .y = valence NOCHILD(1).
//
// Inside, width w, span (i, j=i+w):
//   SL = (+)_r CR(i,i+r).NC + CL(j,i+r+1).HC ;  IL(j,i).v = SL + attach[j,i,v] + dec[j,LEFT ,v,GO]   (dmv.py:50-52)
//   SR = (+)_r CR(i,i+r).HC + CL(j,i+r+1).NC ;  IR(i,j).v = SR + attach[i,j,v] + dec[i,RIGHT,v,GO]   (dmv.py:54-56)
//   CL(j,i).v = (+)_r CL(i+r,i).NC + IL(j,i+r).v                                                     (dmv.py:58-59)
//   CR(i,j).v = (+)_r IR(i,i+1+r).v + CR(i+1+r,j).NC                                                 (dmv.py:61-62)
//   CR(0,w).* = zero unless w == len                                                                 (dmv.py:63)
//
// Work decomposition (both passes): the Ne-w spans of a width are dealt to groups of G lanes
// (G a power of two <= 64, so a group never straddles a wavefront); the lanes of a group split the
// split-point range r = 0..w-1 and combine with a butterfly all-reduce (DPP on the GPU).  Only the
// r = 0 term of CL and the r = w-1 term of CR depend on the incomplete span of the SAME width, and
// that span belongs to the same group -- so ONE barrier per width suffices in each pass.
//
// Outside: every cell is written once, so the charts are the tape; the adjoint of
// out = lse_r t_r is t_r_bar += out_bar * exp(t_r - out) (Max semiring: the first arg-max only).
// Adjoints of complete spans are accumulated in two arrays, gCc (contributions that come from
// complete-span parents; those only ever reach the NOCHILD component, so gCc is a plain float
// chart) and gCi (from incomplete-span parents): within a phase every
// read-modify-write target is then owned by exactly one (span, r) pair -- no atomics, results are
// bit-reproducible (ownership argument: DESIGN.md section 2, HISTORY.md section 2.2).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VLG_HD __device__ __forceinline__
#define VLG_HOSTDEV __host__ __device__ inline
#define VLG_HDM __device__ __forceinline__   // member-function form
#define VLG_HOSTDEV_M __host__ __device__
#define VLG_EXP(x) __builtin_amdgcn_exp2f(x)   // v_exp_f32: 2^x
#define VLG_LOG(x) __builtin_amdgcn_logf(x)    // v_log_f32: log2 x
#define VLG_BITS2F(u) __uint_as_float(u)
#define VLG_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#else
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#define VLG_HD static inline
#define VLG_HOSTDEV static inline
#define VLG_HDM inline
#define VLG_HOSTDEV_M
#define VLG_EXP(x) exp2f(x)
#define VLG_LOG(x) log2f(x)
struct float2 { float x, y; };
static inline float2 make_float2(float a, float b) { float2 r; r.x = a; r.y = b; return r; }
static inline float vlg_bits2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
#define VLG_BITS2F(u) vlg_bits2f(u)
#define VLG_ATOMIC_ADD(p, v) (*(p) += (v))
#endif

#define VLG_NEGINF (-1e12f)  // semiring zero, semirings.py:16,128 (finite sentinel, never -inf)
#define VLG_SR_LOG 0
#define VLG_SR_MAX 1
#define VLG_LOWEST (-3.0e38f)
#if defined(VLG_STAMP) && defined(__HIPCC__)
#define VLG_STAMP_AT(x, k) (x).stamp(k)   // diagnostic build: cycle stamps between the segments of a phase
#else
#define VLG_STAMP_AT(x, k) ((void)0)
#endif
// Charts are kept in log2 units so that the hardware's native 2^x / log2 x need no scaling multiplies:
// potentials are multiplied by log2(e) once at load, logZ by ln(2) once at store; the adjoint weights
// exp(t - out) = 2^((t - out) log2 e) are ratios and need no correction.
#define VLG_LOG2E 1.44269504088896340736f
#define VLG_LN2 0.69314718055994530942f

namespace vlg {

// ---- input element types: potentials arrive as fp32 or bf16; all arithmetic is fp32 ----------------
struct F32In {
    using T = float;
    static VLG_HDM float ld(const T* p, size_t i) { return p[i]; }
    static VLG_HDM float2 ld2(const T* p, size_t i) { return make_float2(p[i], p[i + 1]); }
};
// Potentials may carry -inf (a caller masking with float('-inf'), log_softmax underflow): the reference's logsumexp
// treats it as probability zero.  The DP arithmetic is finite-math (sentinel -1e12, never inf - inf), so potentials are
// clamped to a finite floor as they enter: below every real score, and small enough that sums of a few stay in range.
#define VLG_POT_FLOOR (-1e30f)
VLG_HD float pot_clamp(float x) { return x > VLG_POT_FLOOR ? x : VLG_POT_FLOOR; }   // also maps NaN to the floor
VLG_HD float2 pot_clamp2(float2 x) { return make_float2(pot_clamp(x.x), pot_clamp(x.y)); }
struct BF16In {
    using T = uint16_t;
    static VLG_HDM float ld(const T* p, size_t i) { return VLG_BITS2F((uint32_t)p[i] << 16); }
    static VLG_HDM float2 ld2(const T* p, size_t i) {
        return make_float2(VLG_BITS2F((uint32_t)p[i] << 16), VLG_BITS2F((uint32_t)p[i + 1] << 16));
    }
};

// chart pitch: odd and >= N + 1 so that row-strided (column) walks hit distinct LDS banks
VLG_HOSTDEV int chart_pitch(int N) { return (N + 1) | 1; }

// dec[h] is 8 floats [dir][val][decision]
VLG_HD int dec_idx(int dir, int val, int z) { return (dir * 2 + val) * 2 + z; }

struct DmvCtx {
    int Ne;         // len + 1: only spans inside [0, len] are ever read by valid cells
    int len;
    int P;          // chart pitch (cells), odd, >= N + 1
    float2* C;      // complete spans   [Ne][P]
    float2* I;      // incomplete spans [Ne][P]  (pre-loaded with attach + dec[...,GO])
    float2* C2;     // where the outside pass reads the value charts from: == C, I unless the placement mode overlays the
    float2* I2;     //   adjoint charts on them in LDS (then the inside pass's charts are copied out to the workspace first)
    float* S;       // SL(i,j) at S[j*P+i], SR(i,j) at S[i*P+j]
    float* gCc;     // adjoint of C.NOCHILD, part contributed by complete-span parents (HASCHILD gets none from them)
    float2* gCi;    // adjoint of C, part contributed by incomplete-span parents
    float2* gI;     // adjoint of I (== d logZ / d attach once complete)
    float* decs;    // staged dec        [Ne][8]
    float* gdecs;   // the walk's STOP counts [Ne][dir][val] (Max semiring; empty otherwise: the replay reads them off gCc / gCi)
    unsigned char* bpS;   // Max semiring back-pointers (first arg-max r), same indexing as S
    unsigned char* bpC;   // [Ne][P][2]
    bool walk;            // Max semiring with an outside pass: no S / gCi, gCc is just the walk's stack (lean layout)
};

// weight of term r in a reduction with result `out`, scaled by the upstream adjoint g
template <int SR>
VLG_HD float adj_w(float g, float t, float out, int r, int bp) {
    if (SR == VLG_SR_MAX) return r == bp ? g : 0.f;
    // t <= out up to rounding; the clamp also keeps a zero adjoint zero when `out` is a masked / sentinel cell
    return g * VLG_EXP(fminf(t - out, 0.f));
}

// running max with first-index tie-break (torch.max semantics, semirings.py:199-200)
VLG_HD void upd_max(float& m, int& am, float t, int r) {
    if (t > m) { m = t; am = r; }
}

// fold the single same-width term (value t, index rt) into a partial reduction (m, s, am).
// `t_first`: the extra term precedes every index of the partial (r = 0) -> it wins ties.
template <int SR>
VLG_HD float fold_term(float m, float s, int am, float t, int rt, bool t_first, int& bp) {
    if (SR == VLG_SR_MAX) {
        const bool take = t_first ? (t >= m) : (t > m);
        bp = take ? rt : am;
        return take ? t : m;
    }
    bp = 0;
    // log2(s 2^m + 2^t): one of the two rescales is 2^0, so a single exponential (of -|m - t|) serves both cases
    const float d = m - t, e = VLG_EXP(-fabsf(d));
    const float S = d >= 0.f ? s + e : fmaf(s, e, 1.f);
    return fmaxf(m, t) + VLG_LOG(S);
}

// lanes per span as a power of two: returns log2 G
// The kernel is bound by VALU issue (every wave64 op occupies its SIMD for 4 cycles, v_exp / v_log for 16), so filling
// every lane is not the fastest choice: a wider group means more butterfly steps and more wavefronts per SIMD running
// the same per-span bookkeeping.  The inside pass is fastest when its groups fill about half the block (measured:
// 71.0 -> 61.9 us at B = 256, L = 40); the outside pass, with its longer per-term chains, wants every lane.
#ifndef VLG_DP_LANES_FW
#define VLG_DP_LANES_FW 256
#endif
#ifndef VLG_DP_LANES_BW
#define VLG_DP_LANES_BW 256   // per direction (the two directions of a span run on separate halves of the workgroup); the outside pass has
#endif                        // no cross-lane reduction, so its results do not depend on this (unlike the inside pass's summation trees)
VLG_HD int group_log2(int spans, int w, int nt, int budget) {
    const int cap = budget < nt ? budget : nt;
    int lg = 0;
    while (lg < 6 && (1 << lg) < w && spans * (2 << lg) <= cap) ++lg;
    return lg;
}
#define VLG_GROUP_LOG2_FW(spans, w, nt) group_log2(spans, w, nt, VLG_DP_LANES_FW)
#define VLG_GROUP_LOG2_BW(spans, w, nt) group_log2(spans, w, nt, 512)   // DepTree outside pass: whole workgroup per width

// ------------------------------------------------------------------------------------------------
// Schedule of a pass.  The lane-group size G = 2^lg is non-decreasing in the width (both conditions of group_log2
// relax as w grows and the span count Ne - w shrinks), so a pass is at most seven SEGMENTS of constant lg and the
// width loop runs inside a function that has lg as a template parameter: butterflies are fully unrolled, the
// lane -> (span, split-point residue) mapping and every per-lane address base are computed once per segment, and
// the per-width preamble shrinks to a handful of scalar adds.  first[k] = smallest w with lg(w) >= k:
//   lg(w) >= k  <=>  2^(k-1) < w  and  (Ne - w) 2^k <= cap      (group_log2's loop conditions for l = k-1)
// ------------------------------------------------------------------------------------------------
struct Sched {
    int first[8];   // segment k covers first[k] <= w < first[k+1]; first[0] = 1, first[7] = Ne
};
VLG_HD Sched make_sched(int Ne, int lanes, int budget) {
    const int cap = budget < lanes ? budget : lanes;
    Sched s;
    s.first[0] = 1;
    for (int k = 1; k <= 6; ++k) {
        int a = (1 << (k - 1)) + 1, b = Ne - (cap >> k);
        int f = a > b ? a : b;
        if (f < s.first[k - 1]) f = s.first[k - 1];
        s.first[k] = f < Ne ? f : Ne;
    }
    s.first[7] = Ne;
    return s;
}

// ---- the short-sentence code image (round 5) ---------------------------------------------------------------------------------------
// A lane holds T = ceil(w / G) split points of a span; the span bodies are unrolled for T = 1 ... 4 with generic loops behind them, per
// segment and direction: ~70 + 35 bodies, of which a sentence of <= 40 words executes 18 + 9.  The bodies it never executes still cost:
// the fused headline launch measured 75.9 us with them compiled in and 69.7 us without (same bits) -- the code image, not the
// arithmetic.  For N <= kShortN every width has T <= 3 (checked at compile time below for the lane budgets in force), so the kernels
// are instantiated once more with the span-count policy kSpansShort, whose dispatch ends at T = 3, and the host launches that
// instantiation for N <= kShortN.  Span-count policy (the LONGSPAN template parameter): 0 general, 1 long sentences (chunked long
// spans, workspace placements), 2 short sentences.
constexpr int kSpansGeneral = 0, kSpansLong = 1, kSpansShort = 2;
constexpr int kShortN = 41;
constexpr int group_log2_c(int spans, int w, int cap) {
    int lg = 0;
    while (lg < 6 && (1 << lg) < w && spans * (2 << lg) <= cap) ++lg;
    return lg;
}
constexpr bool short_sentence_ok(int n_max, int cap) {
    for (int ne = 2; ne <= n_max; ++ne)
        for (int w = 1; w < ne; ++w) {
            const int lg = group_log2_c(ne - w, w, cap);
            if (((w + (1 << lg) - 1) >> lg) > 3) return false;
        }
    return true;
}
// the largest T that a segment of group size 2^lg meets in ANY sentence of N <= n_max: the short image's dispatch of that segment ends there
// (N <= 41 at 256 lanes per direction: lg 0, 1, 6 -> 1; lg 2, 4, 5 -> 2; lg 3 -> 3: 12 of the 21 bodies per direction are compiled)
constexpr int short_tmax(int lg_want, int n_max, int cap) {
    int m = 1;
    for (int ne = 2; ne <= n_max; ++ne)
        for (int w = 1; w < ne; ++w) {
            const int lg = group_log2_c(ne - w, w, cap), t = (w + (1 << lg) - 1) >> lg;
            if (lg == lg_want && t > m) m = t;
        }
    return m;
}
static_assert(short_sentence_ok(kShortN, VLG_DP_LANES_FW) && short_sentence_ok(kShortN, VLG_DP_LANES_BW),
              "kShortN: a sentence of that length has a width with more than three split points per lane at these lane budgets");

#if defined(__HIPCC__)
#define VLG_MUL24(a, b) __mul24((a), (b))   // v_mul_i32_i24: full rate (v_mul_lo_u32 is quarter rate); all chart indices < 2^17
#else
#define VLG_MUL24(a, b) ((a) * (b))
#endif

// ------------------------------------------------------------------------------------------------
// DMV1o inside, width w, ONE span (i, j = i + w), ONE direction, handled by a group of G = 2^LG lanes (this lane covers
// split points r = rr, rr+G, ...).  The left-facing chain SL -> IL(j,i) -> CL(j,i) and the right-facing chain
// SR -> IR(i,j) -> CR(i,j) of a span do not depend on each other within a width, so the two halves of the workgroup
// take one each (DIR 0 / 1): three reductions per lane instead of six, on twice the wavefronts.
// TU > 0: exactly TU iterations per lane, the terms of every iteration stay in registers between the max pass and the
// sum pass (no LDS re-read, no branches: out-of-range r is clamped for the loads and masked to the lowest float).
// TU == 0: generic loops for long spans.  D = i (P + 1) is the chart index of the diagonal cell of row i.
//   DIR 0 reductions: 0 SL, 1 CL.x, 2 CL.y (r >= 1)        DIR 1: 0 SR, 1 CR.x, 2 CR.y (r <= w-2)
// ------------------------------------------------------------------------------------------------
// NOCLAMP (device, every chart in LDS): the loads of a lane's out-of-range split points are NOT clamped to the last valid one -- an
// LDS read cannot fault (past the allocation it returns 0), its value is masked to the lowest float like the clamped duplicate was,
// and without the clamp the addresses of a lane's TU terms are one base plus compile-time multiples of G: immediate offsets instead
// of a min, a multiply and a shift-add per term.  (The host emulator's charts are heap arrays: it keeps the clamp.)
template <int SR, bool BWD, int DIRT, int TU, typename X, bool NOCLAMP = false>
VLG_HD void dmv_fw_span(const DmvCtx& c, int w, int G, int D, bool live, int rr, X& x, int dir_rt = 0) {
    const int DIR = DIRT < 0 ? dir_rt : DIRT;   // DIRT < 0: the direction is a (wave-uniform) run-time value -- both wave halves run ONE copy of the code
    VLG_STAMP_AT(x, 6);   // since the end of the previous span: phase preamble + barrier
    const int P = c.P, DW = D + VLG_MUL24(w, P);   // DW: chart index of (row j, column i)
    const float* Cf = reinterpret_cast<const float*>(c.C);
    // element indices at r = 0 (float units for the single-component reads)
    const int eA = 2 * (D + 1) + (DIR == 0 ? 1 : 0);          // CR(i, i+r):   .NC for SL, .HC for SR
    const int eB = 2 * (DW + 1) + (DIR == 0 ? 0 : 1);         // CL(j, i+r+1): .HC for SL, .NC for SR
    const int eU = DIR == 0 ? 2 * D + 1 : 2 * (D + P + w + 1) + 1;   // CL(i+r, i).NC  |  CR(i+1+r, j).NC   (stride P)
    const int eV = DIR == 0 ? DW : D + 2;                     // IL(j, i+r)     |  IR(i, i+1+r)
    const int kO = DIR == 0 ? DW : D + w + 1;                 // this span's own slot: IL(j,i) / CL(j,i)  |  IR(i,j) / CR(i,j)
    // per-span constants first: their LDS latency hides under the reductions
    const float2 aX = c.I[kO];                                // attach + dec[...,GO], staged at load
    const float cX = DIR == 0 ? Cf[2 * D + 1] : Cf[2 * (DW + w + 1) + 1];   // CL(i,i).NC | CR(j,j).NC: partner of the same-width term
    float m[3], s[3] = {0.f, 0.f, 0.f};
    int am[3];
    VLG_STAMP_AT(x, 7);   // pointer set-up + per-span constant loads
    if (TU > 0) {
        float t[TU > 0 ? TU : 1][3];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int r = rr + u * G, rc = NOCLAMP ? r : (r < w ? r : w - 1);
#ifdef VLG_ABL_NOLOADS
            const float a = (float)rc, b = aX.x, uu = cX;
            const float2 vv = aX;
#else
            const float a = Cf[eA + 2 * rc], b = Cf[eB + 2 * rc];
            const float uu = Cf[eU + 2 * VLG_MUL24(rc, P)];
            const float2 vv = c.I[eV + rc];
#endif
            const bool v0 = r < w, v1 = DIR == 0 ? (v0 && r >= 1) : (r <= w - 2);
            t[u][0] = v0 ? a + b : VLG_LOWEST;
            t[u][1] = v1 ? uu + vv.x : VLG_LOWEST;
            t[u][2] = v1 ? uu + vv.y : VLG_LOWEST;
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            m[k] = t[0][k];
            am[k] = rr;
            if (SR == VLG_SR_LOG) {
                // (no position wanted: a plain v_max.  The compare / select form below stays compare + select in the Log
                //  instantiation too -- hipcc may not fold it without -fno-signed-zeros -- and each pair waits two states on VCC)
#pragma unroll
                for (int u = 1; u < TU; ++u) m[k] = fmaxf(m[k], t[u][k]);
            } else {
#pragma unroll
                for (int u = 1; u < TU; ++u)
                    if (t[u][k] > m[k]) { m[k] = t[u][k]; am[k] = rr + u * G; }   // strict: first index wins ties
            }
        }
        VLG_STAMP_AT(x, 1);
#ifndef VLG_ABL_NOMAXBFLY
        if (SR == VLG_SR_MAX) x.template allreduce_argmax<3>(m, am, G);
        else x.template allreduce_max<3>(m, G);
#endif
        VLG_STAMP_AT(x, 2);
        if (SR == VLG_SR_LOG) {
#ifdef VLG_ABL_NOEXP
            for (int k = 0; k < 3; ++k) s[k] = t[0][k] - m[k];
#else
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                s[k] = VLG_EXP(t[0][k] - m[k]);                                  // (= 0 + the first term, bit for bit)
#pragma unroll
                for (int u = 1; u < TU; ++u) s[k] += VLG_EXP(t[u][k] - m[k]);   // masked terms: 2^(-3e38) = 0
            }
#endif
            VLG_STAMP_AT(x, 3);
#ifndef VLG_ABL_NOSUMBFLY
            x.template allreduce_sum<3>(s, G);
#endif
        }
        VLG_STAMP_AT(x, 4);
    } else if (TU < 0) {
        // long spans in the long-sentence placements (TU < 0): the same two passes, four split points at a time -- the loads of
        // a chunk are independent and branch-free (out-of-range r clamped for the load, masked to the lowest float), where a
        // loop over single split points with its `if` pays an LDS round trip per split point.  Same operations in the same
        // order as that loop: identical bits.
        constexpr int CU = 4;
        auto terms = [&](int r0, float (*t)[3]) {
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int r = r0 + u * G, rc = r < w ? r : w - 1;
                const float a = Cf[eA + 2 * rc], b = Cf[eB + 2 * rc];
                const float uu = Cf[eU + 2 * VLG_MUL24(rc, P)];
                const float2 vv = c.I[eV + rc];
                const bool v0 = r < w, v1 = DIR == 0 ? (v0 && r >= 1) : (r <= w - 2);
                t[u][0] = v0 ? a + b : VLG_LOWEST;
                t[u][1] = v1 ? uu + vv.x : VLG_LOWEST;
                t[u][2] = v1 ? uu + vv.y : VLG_LOWEST;
            }
        };
        for (int k = 0; k < 3; ++k) { m[k] = VLG_LOWEST; am[k] = 0; }
        for (int r0 = rr; r0 < w; r0 += CU * G) {
            float t[CU][3];
            terms(r0, t);
#pragma unroll
            for (int u = 0; u < CU; ++u)
#pragma unroll
                for (int k = 0; k < 3; ++k) upd_max(m[k], am[k], t[u][k], r0 + u * G);
        }
        if (SR == VLG_SR_MAX) x.template allreduce_argmax<3>(m, am, G);
        else x.template allreduce_max<3>(m, G);
        if (SR == VLG_SR_LOG) {
            for (int r0 = rr; r0 < w; r0 += CU * G) {
                float t[CU][3];
                terms(r0, t);
#pragma unroll
                for (int u = 0; u < CU; ++u)
#pragma unroll
                    for (int k = 0; k < 3; ++k) s[k] += VLG_EXP(t[u][k] - m[k]);   // masked terms: 2^(-3e38) = 0
            }
            x.template allreduce_sum<3>(s, G);
        }
    } else {
        // generic loop over single split points (placements with every chart in LDS, where N <= 61 rarely gets here: the chunked
        // form above costs the all-in-LDS kernel registers and the headline 0.8 %)
        for (int k = 0; k < 3; ++k) { m[k] = VLG_LOWEST; am[k] = 0; }
        for (int r = rr; r < w; r += G) {
            upd_max(m[0], am[0], Cf[eA + 2 * r] + Cf[eB + 2 * r], r);
            if (DIR == 0 ? r >= 1 : r <= w - 2) {
                const float uu = Cf[eU + 2 * VLG_MUL24(r, P)];
                const float2 vv = c.I[eV + r];
                upd_max(m[1], am[1], uu + vv.x, r);
                upd_max(m[2], am[2], uu + vv.y, r);
            }
        }
        if (SR == VLG_SR_MAX) x.template allreduce_argmax<3>(m, am, G);
        else x.template allreduce_max<3>(m, G);
        if (SR == VLG_SR_LOG) {
            for (int r = rr; r < w; r += G) {
                s[0] += VLG_EXP(Cf[eA + 2 * r] + Cf[eB + 2 * r] - m[0]);
                if (DIR == 0 ? r >= 1 : r <= w - 2) {
                    const float uu = Cf[eU + 2 * VLG_MUL24(r, P)];
                    const float2 vv = c.I[eV + r];
                    s[1] += VLG_EXP(uu + vv.x - m[1]);
                    s[2] += VLG_EXP(uu + vv.y - m[2]);
                }
            }
            x.template allreduce_sum<3>(s, G);
        }
    }
#ifdef VLG_ABL_NOEPI   // timing ablation (wrong results): no log / fold after the butterflies
    const float Sv = m[0] + s[0];
    const float2 In = make_float2(aX.x + Sv, aX.y + Sv);
    int b0 = 0, b1 = 0;
    float Cx = m[1] + s[1] + cX, Cy = m[2] + s[2] + cX;
#else
    const float Sv = SR == VLG_SR_LOG ? m[0] + VLG_LOG(s[0]) : m[0];       // SL | SR
    const float2 In = make_float2(aX.x + Sv, aX.y + Sv);                    // IL(j,i) | IR(i,j)
    int b0, b1;
    // the same-width term: r = 0 of CL(j,i) (it precedes every other index -> wins ties), r = w-1 of CR(i,j)
    float Cx = fold_term<SR>(m[1], s[1], am[1], cX + In.x, DIR == 0 ? 0 : w - 1, DIR == 0, b0);
    float Cy = fold_term<SR>(m[2], s[2], am[2], cX + In.y, DIR == 0 ? 0 : w - 1, DIR == 0, b1);
#endif
    if (DIR == 1 && D == 0 && w != c.len) { Cx = VLG_NEGINF; Cy = VLG_NEGINF; }   // single root, dmv.py:63  (D == 0 <=> i == 0)
    VLG_STAMP_AT(x, 5);
    if (live && rr == 0) {
        c.I[kO] = In;
        c.C[kO] = make_float2(Cx, Cy);
        if (BWD) {
            const int kS = DIR == 0 ? DW : D + w;      // SL(i,j) at S[j*P+i], SR(i,j) at S[i*P+j]
            if (SR != VLG_SR_MAX) c.S[kS] = Sv;        // the tape of the outside replay; the Max semiring's back-pointer walk does not read it
            if (SR == VLG_SR_MAX) {
                c.bpS[kS] = (unsigned char)am[0];
                c.bpC[kO * 2] = (unsigned char)b0;
                c.bpC[kO * 2 + 1] = (unsigned char)b1;
            }
        }
    }
    VLG_STAMP_AT(x, 8);   // stores
}

// one width of the inside pass for this lane: G = 2^LG lanes per (span, direction); LG < 0: G is a run-time value
template <int SR, bool BWD, int DIR, int LG, int LONGSPAN, typename X>
VLG_HD void dmv_fw_width(const DmvCtx& c, int w, int lgr, int t, int nd, X& x, int dir_rt = 0) {
    const int lg = LG >= 0 ? LG : lgr, G = 1 << lg, per = nd >> lg;
    const int rr = t & (G - 1), slot = t >> lg, spans = c.Ne - w;
    const int T = (w + G - 1) >> lg;   // split points per lane; uniform over the workgroup
    for (int base = 0; base < spans; base += per) {
        const bool live = base + slot < spans;
        const int i = live ? base + slot : 0;   // dead lanes shadow span 0 and never store
        if (X::kSkipDeadWaves && (base + ((t & ~63) >> lg)) >= spans) continue;   // whole wavefront past the last span
        const int D = VLG_MUL24(i, c.P + 1);
        constexpr bool NC = LONGSPAN != 1 && X::kChartsInLds;
        if constexpr (LONGSPAN == kSpansShort && LG >= 0) {   // N <= kShortN: this segment never sees more than TM split points per lane
            constexpr int TM = short_tmax(LG, kShortN, VLG_DP_LANES_FW);
            if (TM == 1 || T == 1) dmv_fw_span<SR, BWD, DIR, 1, X, NC>(c, w, G, D, live, rr, x, dir_rt);
            else if (TM == 2 || T == 2) dmv_fw_span<SR, BWD, DIR, 2, X, NC>(c, w, G, D, live, rr, x, dir_rt);
            else dmv_fw_span<SR, BWD, DIR, 3, X, NC>(c, w, G, D, live, rr, x, dir_rt);
        } else if (T == 1) dmv_fw_span<SR, BWD, DIR, 1, X, NC>(c, w, G, D, live, rr, x, dir_rt);
        else if (T == 2) dmv_fw_span<SR, BWD, DIR, 2, X, NC>(c, w, G, D, live, rr, x, dir_rt);
        else if (T == 3) dmv_fw_span<SR, BWD, DIR, 3, X, NC>(c, w, G, D, live, rr, x, dir_rt);
        else if (T == 4) dmv_fw_span<SR, BWD, DIR, 4, X, NC>(c, w, G, D, live, rr, x, dir_rt);
        else dmv_fw_span<SR, BWD, DIR, LONGSPAN == kSpansLong ? -1 : 0>(c, w, G, D, live, rr, x, dir_rt);
    }
}

// all widths of one segment (constant group size), one barrier per width
template <int SR, bool BWD, int LG, int LONGSPAN, typename X>
VLG_HD void dmv_fw_segment(const DmvCtx& c, int w0, int w1, int tid, int nt, X& x) {
    const int nd = nt >> 1;                 // lanes per direction
    const bool right = x.uniform(tid >= nd);   // wave-uniform on the device (nd is a multiple of 64)
    const int t = right ? tid - nd : tid;
    for (int w = w0; w < w1; ++w) {
#ifndef VLG_ABL_NOBODY
#ifdef VLG_DIR_RT_FW   // (measured: 33.4 -> 39.4 us for the inside pass -- its selects sit in the per-term loop; see dmv_bw_segment)
        dmv_fw_width<SR, BWD, -1, LG, LONGSPAN>(c, w, LG, t, nd, x, right ? 1 : 0);
#else
        if (right) dmv_fw_width<SR, BWD, 1, LG, LONGSPAN>(c, w, LG, t, nd, x);
        else dmv_fw_width<SR, BWD, 0, LG, LONGSPAN>(c, w, LG, t, nd, x);
#endif
#endif
#ifndef VLG_ABL_NOBARRIER
        x.sync();
#endif
    }
}

template <int SR, bool BWD, int LONGSPAN, typename X>
VLG_HD void dmv_fw_all(const DmvCtx& c, int tid, int nt, X& x) {
    const Sched sc = make_sched(c.Ne, nt >> 1, VLG_DP_LANES_FW);
    dmv_fw_segment<SR, BWD, 0, LONGSPAN>(c, sc.first[0], sc.first[1], tid, nt, x);
    dmv_fw_segment<SR, BWD, 1, LONGSPAN>(c, sc.first[1], sc.first[2], tid, nt, x);
    dmv_fw_segment<SR, BWD, 2, LONGSPAN>(c, sc.first[2], sc.first[3], tid, nt, x);
    dmv_fw_segment<SR, BWD, 3, LONGSPAN>(c, sc.first[3], sc.first[4], tid, nt, x);
    dmv_fw_segment<SR, BWD, 4, LONGSPAN>(c, sc.first[4], sc.first[5], tid, nt, x);
    dmv_fw_segment<SR, BWD, 5, LONGSPAN>(c, sc.first[5], sc.first[6], tid, nt, x);
    dmv_fw_segment<SR, BWD, 6, LONGSPAN>(c, sc.first[6], sc.first[7], tid, nt, x);
}

// ------------------------------------------------------------------------------------------------
// DMV1o outside, width w, ONE span, ONE direction (DIR 0: the adjoints of CL(j,i), then of IL(j,i) through SL;
// DIR 1: CR(i,j), then IR(i,j) through SR).  The two directions of a span touch disjoint adjoint words -- the
// left one owns gI row j, the CL half of gCc, and the (.NC of CR(i,.), .HC of CL(j,.)) components of gCi; the right
// one the mirror set -- so, like the inside pass, they run on separate halves of the workgroup.
// All loads -- including the read half of every read-modify-write -- are issued before the first store, so
// nothing waits on an earlier store of the same phase.
// ------------------------------------------------------------------------------------------------
template <int SR, int DIRT, int TU, typename X, bool CHUNKED = false, bool NOCLAMP = false>   // NOCLAMP: see dmv_fw_span
VLG_HD void dmv_bw_span(const DmvCtx& c, int w, int G, int D, bool live, int rr, X& x, int nchunk = 1, int dir_rt = 0) {
    const int DIR = DIRT < 0 ? dir_rt : DIRT;
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    const float* Cf = reinterpret_cast<const float*>(c.C);
    float* gCif = reinterpret_cast<float*>(c.gCi);
    const int eA = 2 * (D + 1) + (DIR == 0 ? 1 : 0);          // CR(i, i+r):   .NC for SL, .HC for SR   (also its gCi word)
    const int eB = 2 * (DW + 1) + (DIR == 0 ? 0 : 1);         // CL(j, i+r+1): .HC for SL, .NC for SR
    const int eU = DIR == 0 ? D : D + P + w + 1;              // cell of CL(i+r, i) | CR(i+1+r, j)   (stride P): value .NC, adjoint gCc
    const int eV = DIR == 0 ? DW : D + 2;                     // IL(j, i+r) | IR(i, i+1+r)
    const int kO = DIR == 0 ? DW : D + w + 1;                 // this span's own slot
    const int kS = DIR == 0 ? DW : D + w;                     // SL(i,j) at S[j*P+i], SR(i,j) at S[i*P+j]
    const int selfr = DIR == 0 ? 0 : w - 1;                   // the split point whose incomplete span has this same width
    float2 gc = make_float2(0.f, 0.f);                        // total adjoint of CL(j,i) | CR(i,j)
    {
        const float a = c.gCc[kO];
        const float2 b = c.gCi[kO];
        if (live && !(DIR == 1 && D == 0 && w != c.len)) gc = make_float2(b.x, a + b.y);   // masked root cell: dmv.py:63
    }
    const float2 oc = c.C[kO], gi_old = c.gI[kO];
    const float Sv = c.S[kS];
    int b0 = 0, b1 = 0, bs = 0;
    if (SR == VLG_SR_MAX) { b0 = c.bpC[kO * 2]; b1 = c.bpC[kO * 2 + 1]; bs = c.bpS[kS]; }
    // share of gI(own slot) that comes from this very span's complete cell: the r = 0 (left) / r = w-1 (right) term of its
    // reduction.  Every lane of the group evaluates it (two same-address LDS reads, two weights): cheaper than handing it
    // round from the one lane whose split-point range contains it (a ds_bpermute round trip in the middle of the phase).
    float self[2];
    const float su = Cf[2 * (eU + VLG_MUL24(selfr, P)) + 1];   // CL(i,i).NC | CR(j,j).NC
    const float2 sv = c.I[eV + selfr];                          // IL(j,i) | IR(i,j): this span's own incomplete value
    if (TU > 0) {
        // nchunk > 1 (long sentences: more than 4 split points per lane): the split-point range is walked in chunks of TU per
        // lane, each chunk with all of its loads ahead of its stores -- one memory round trip per chunk, where a loop over
        // single split points pays one per split point (with the value charts in the workspace that round trip is an L2
        // access: the generic loop below made the outside pass at L = 80 twice as long as the inside pass).
        float2 gi = make_float2(0.f, 0.f);
        float gs = 0.f;
        if (CHUNKED) {   // needed by every chunk: ahead of the loop (the one-chunk instantiation keeps it behind its loads, below)
            self[0] = adj_w<SR>(gc.x, su + sv.x, oc.x, selfr, b0);
            self[1] = adj_w<SR>(gc.y, su + sv.y, oc.y, selfr, b1);
            gi = make_float2(gi_old.x + self[0], gi_old.y + self[1]);   // complete adjoint of IL(j,i) | IR(i,j)
            gs = gi.x + gi.y;
        }
        for (int ch = 0; ch < (CHUNKED ? nchunk : 1); ++ch) {
            const int r0 = rr + ch * (TU * G);
            float uu[TU > 0 ? TU : 1], xa[TU > 0 ? TU : 1], xb[TU > 0 ? TU : 1], o_c[TU > 0 ? TU : 1], o_ga[TU > 0 ? TU : 1],
                o_gb[TU > 0 ? TU : 1], w0[TU > 0 ? TU : 1], w1[TU > 0 ? TU : 1];
            float2 vv[TU > 0 ? TU : 1], o_gi[TU > 0 ? TU : 1];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int r = r0 + u * G, rc = NOCLAMP ? r : (r < w ? r : w - 1), rP = VLG_MUL24(rc, P);
                uu[u] = Cf[2 * (eU + rP) + 1];
                vv[u] = c.I[eV + rc];
                xa[u] = Cf[eA + 2 * rc];
                xb[u] = Cf[eB + 2 * rc];
#ifdef VLG_ABL_BW_NOADJLOAD   // timing ablation (wrong results): what the four adjoint loads per split point cost
                o_gi[u] = vv[u]; o_c[u] = uu[u]; o_ga[u] = xa[u]; o_gb[u] = xb[u];
#else
                o_gi[u] = c.gI[eV + rc];
                o_c[u] = c.gCc[eU + rP];
                o_ga[u] = gCif[eA + 2 * rc];
                o_gb[u] = gCif[eB + 2 * rc];
#endif
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int r = r0 + u * G;
                const bool ok = r < w;
                w0[u] = ok ? adj_w<SR>(gc.x, uu[u] + vv[u].x, oc.x, r, b0) : 0.f;
                w1[u] = ok ? adj_w<SR>(gc.y, uu[u] + vv[u].y, oc.y, r, b1) : 0.f;
            }
            if (!CHUNKED) {
                self[0] = adj_w<SR>(gc.x, su + sv.x, oc.x, selfr, b0);
                self[1] = adj_w<SR>(gc.y, su + sv.y, oc.y, selfr, b1);
            }
            x.lockstep();   // the lanes of a group sit in one wavefront: every load above precedes every store below
            if (!CHUNKED) {
                gi = make_float2(gi_old.x + self[0], gi_old.y + self[1]);
                gs = gi.x + gi.y;
            }
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int r = r0 + u * G;
                if (live && r < w) {
                    const float ws = adj_w<SR>(gs, xa[u] + xb[u], Sv, r, bs);
                    if (r != selfr) c.gI[eV + r] = make_float2(o_gi[u].x + w0[u], o_gi[u].y + w1[u]);
                    c.gCc[eU + VLG_MUL24(r, P)] = o_c[u] + (w0[u] + w1[u]);
                    gCif[eA + 2 * r] = o_ga[u] + ws;
                    gCif[eB + 2 * r] = o_gb[u] + ws;
                }
            }
        }
        if (live && rr == 0) c.gI[kO] = gi;   // == d logZ / d attach[j,i,:] | attach[i,j,:]
        return;
    }
    for (int r = rr; r < w; r += G) {
        const int rP = VLG_MUL24(r, P);
        const float a = Cf[2 * (eU + rP) + 1];
        const float2 v = c.I[eV + r];
        const float q0 = adj_w<SR>(gc.x, a + v.x, oc.x, r, b0);
        const float q1 = adj_w<SR>(gc.y, a + v.y, oc.y, r, b1);
        if (r != selfr && live) { const float2 t = c.gI[eV + r]; c.gI[eV + r] = make_float2(t.x + q0, t.y + q1); }
        if (live) c.gCc[eU + rP] += q0 + q1;
    }
    self[0] = adj_w<SR>(gc.x, su + sv.x, oc.x, selfr, b0);
    self[1] = adj_w<SR>(gc.y, su + sv.y, oc.y, selfr, b1);
    x.lockstep();
    const float2 gi = make_float2(gi_old.x + self[0], gi_old.y + self[1]);
    const float gs = gi.x + gi.y;
    for (int r = rr; r < w; r += G) {
        const float ws = adj_w<SR>(gs, Cf[eA + 2 * r] + Cf[eB + 2 * r], Sv, r, bs);
        if (live) {
            gCif[eA + 2 * r] += ws;
            gCif[eB + 2 * r] += ws;
        }
    }
    if (live && rr == 0) c.gI[kO] = gi;
}

// ---- Outside pass, short-sentence image: the VALUE reads of a width issued one width ahead (round 6) -------------------------------
// A span body of the outside pass reads (a) inside-pass results -- its own C / S cells, the same-width term, and per split point the
// four value cells the weights exp(t - out) are made of: nothing writes them during this pass -- and (b) adjoint cells: its own
// total (final only after the previous width's barrier) and the read-modify-write targets.  With everything requested behind the
// barrier, the (a) reads queue in front of the (b) reads of all eight wavefronts and the exponentials wait for both.  Here the (a)
// reads of width w - 1 are issued inside width w's body -- behind its adjoint reads, ahead of its stores, so they are complete at the
// barrier like everything else and cost no wait of their own -- and stay in registers across the barrier: behind it only the
// adjoint reads are requested, and the exponentials of the weights run while those are in flight.  Same operations on the same
// operands in the same order per result: bit-identical counts (tools/time_headline.py's SHA-256).
template <int TM>
struct BwVals {
    float2 oc, sv, vv[TM];
    float Sv, su, uu[TM], xa[TM], xb[TM];
};

template <int DIRT, int TM, bool NOCLAMP>
VLG_HD void dmv_bw_load_vals(const DmvCtx& c, int w, int G, int D, int rr, BwVals<TM>& v, int dir_rt) {
    const int DIR = DIRT < 0 ? dir_rt : DIRT;
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    const float* Cf = reinterpret_cast<const float*>(c.C);
    const int eA = 2 * (D + 1) + (DIR == 0 ? 1 : 0), eB = 2 * (DW + 1) + (DIR == 0 ? 0 : 1);
    const int eU = DIR == 0 ? D : D + P + w + 1, eV = DIR == 0 ? DW : D + 2;
    const int kO = DIR == 0 ? DW : D + w + 1, kS = DIR == 0 ? DW : D + w;
    const int selfr = DIR == 0 ? 0 : w - 1;
    v.oc = c.C[kO];
    v.Sv = c.S[kS];
    v.su = Cf[2 * (eU + VLG_MUL24(selfr, P)) + 1];
    v.sv = c.I[eV + selfr];
#pragma unroll
    for (int u = 0; u < TM; ++u) {
        const int r = rr + u * G, rc = NOCLAMP ? r : (r < w ? r : w - 1), rP = VLG_MUL24(rc, P);
        v.uu[u] = Cf[2 * (eU + rP) + 1];
        v.vv[u] = c.I[eV + rc];
        v.xa[u] = Cf[eA + 2 * rc];
        v.xb[u] = Cf[eB + 2 * rc];
    }
}

// dmv_bw_span (TU > 0, one chunk) on prefetched values `v`; with `more`, `nxt` receives the values of width wn for the span at Dn
template <int SR, int DIRT, int TU, int TM, typename X, bool NOCLAMP>
VLG_HD void dmv_bw_span_p(const DmvCtx& c, int w, int G, int D, bool live, int rr, X& x, const BwVals<TM>& v, BwVals<TM>& nxt, bool more,
                          int wn, int Dn, int dir_rt) {
    const int DIR = DIRT < 0 ? dir_rt : DIRT;
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    float* gCif = reinterpret_cast<float*>(c.gCi);
    const int eA = 2 * (D + 1) + (DIR == 0 ? 1 : 0), eB = 2 * (DW + 1) + (DIR == 0 ? 0 : 1);
    const int eU = DIR == 0 ? D : D + P + w + 1, eV = DIR == 0 ? DW : D + 2;
    const int kO = DIR == 0 ? DW : D + w + 1, kS = DIR == 0 ? DW : D + w;
    const int selfr = DIR == 0 ? 0 : w - 1;
    // ---- adjoint reads: what the barrier was for ----
    const float ga = c.gCc[kO];
    const float2 gb = c.gCi[kO];
    const float2 gi_old = c.gI[kO];
    int b0 = 0, b1 = 0, bs = 0;
    if (SR == VLG_SR_MAX) { b0 = c.bpC[kO * 2]; b1 = c.bpC[kO * 2 + 1]; bs = c.bpS[kS]; }
    float o_c[TU], o_ga[TU], o_gb[TU];
    float2 o_gi[TU];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
        const int r = rr + u * G, rc = NOCLAMP ? r : (r < w ? r : w - 1), rP = VLG_MUL24(rc, P);
        o_gi[u] = c.gI[eV + rc];
        o_c[u] = c.gCc[eU + rP];
        o_ga[u] = gCif[eA + 2 * rc];
        o_gb[u] = gCif[eB + 2 * rc];
    }
    // ---- the weights' exponentials, from registers, while those reads are in flight (adj_w = g * 2^min(t - out, 0): the factor first) ----
    float e0[TU], e1[TU], es[TU], s0 = 0.f, s1 = 0.f;
    if (SR != VLG_SR_MAX) {
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            e0[u] = VLG_EXP(fminf(v.uu[u] + v.vv[u].x - v.oc.x, 0.f));
            e1[u] = VLG_EXP(fminf(v.uu[u] + v.vv[u].y - v.oc.y, 0.f));
            es[u] = VLG_EXP(fminf(v.xa[u] + v.xb[u] - v.Sv, 0.f));
        }
        s0 = VLG_EXP(fminf(v.su + v.sv.x - v.oc.x, 0.f));
        s1 = VLG_EXP(fminf(v.su + v.sv.y - v.oc.y, 0.f));
    }
    float2 gc = make_float2(0.f, 0.f);                        // total adjoint of CL(j,i) | CR(i,j)
    if (live && !(DIR == 1 && D == 0 && w != c.len)) gc = make_float2(gb.x, ga + gb.y);   // masked root cell: dmv.py:63
    float w0[TU], w1[TU], self[2];
#pragma unroll
    for (int u = 0; u < TU; ++u) {
        const int r = rr + u * G;
        const bool ok = r < w;
        if (SR == VLG_SR_MAX) {
            w0[u] = ok ? adj_w<SR>(gc.x, 0.f, 0.f, r, b0) : 0.f;
            w1[u] = ok ? adj_w<SR>(gc.y, 0.f, 0.f, r, b1) : 0.f;
        } else {
            w0[u] = ok ? gc.x * e0[u] : 0.f;
            w1[u] = ok ? gc.y * e1[u] : 0.f;
        }
    }
    if (SR == VLG_SR_MAX) {
        self[0] = adj_w<SR>(gc.x, 0.f, 0.f, selfr, b0);
        self[1] = adj_w<SR>(gc.y, 0.f, 0.f, selfr, b1);
    } else {
        self[0] = gc.x * s0;
        self[1] = gc.y * s1;
    }
    // ---- the next width's value reads: ahead of this width's stores ----
    if (more) dmv_bw_load_vals<DIRT, TM, NOCLAMP>(c, wn, G, Dn, rr, nxt, dir_rt);
    x.lockstep();   // the lanes of a group sit in one wavefront: every load above precedes every store below
    const float2 gi = make_float2(gi_old.x + self[0], gi_old.y + self[1]);   // complete adjoint of IL(j,i) | IR(i,j)
    const float gs = gi.x + gi.y;
#pragma unroll
    for (int u = 0; u < TU; ++u) {
        const int r = rr + u * G;
        if (live && r < w) {
            const float ws = SR == VLG_SR_MAX ? adj_w<SR>(gs, 0.f, 0.f, r, bs) : gs * es[u];
            if (r != selfr) c.gI[eV + r] = make_float2(o_gi[u].x + w0[u], o_gi[u].y + w1[u]);
            c.gCc[eU + VLG_MUL24(r, P)] = o_c[u] + (w0[u] + w1[u]);
            gCif[eA + 2 * r] = o_ga[u] + ws;
            gCif[eB + 2 * r] = o_gb[u] + ws;
        }
    }
    if (live && rr == 0) c.gI[kO] = gi;   // == d logZ / d attach[j,i,:] | attach[i,j,:]
}

// widths w1-1 ... w0 of one segment of the short-sentence image (one span per lane group and width: (Ne - w) G <= the lanes of a direction)
template <int SR, int LG, typename X>
VLG_HD void dmv_bw_segment_p(const DmvCtx& c, int w0, int w1, int tid, int nt, X& x) {
    constexpr int G = 1 << LG;
    constexpr int TM = short_tmax(LG, kShortN, VLG_DP_LANES_BW);
    constexpr bool NC = X::kChartsInLds;
    if (w1 <= w0) return;
    const int nd = nt >> 1;
    const bool right = x.uniform(tid >= nd);
    const int dir = right ? 1 : 0;
    const int t = right ? tid - nd : tid;
    const int rr = t & (G - 1), slot = t >> LG, wslot = (t & ~63) >> LG;   // wslot: the first span of this lane's wavefront
    const int Ds = VLG_MUL24(slot, c.P + 1);
    BwVals<TM> va, vb;   // ping-pong: the body of one width reads one and fills the other (no copies)
    {
        const int spans = c.Ne - (w1 - 1);
        dmv_bw_load_vals<-1, TM, NC>(c, w1 - 1, G, slot < spans ? Ds : 0, rr, va, dir);
    }
    auto one_width = [&](int w, const BwVals<TM>& cur, BwVals<TM>& nxt) {
        const int spans = c.Ne - w;
        const bool live = slot < spans;
        const int D = live ? Ds : 0;
        const bool more = w > w0;
        const int Dn = slot < spans + 1 ? Ds : 0;   // the span of width w - 1 (one more span than this width has)
        const int T = (w + G - 1) >> LG;
        if (X::kSkipDeadWaves && wslot >= spans) {   // (wave-uniform) no span of this width in this wavefront: only the prefetch
            if (more) dmv_bw_load_vals<-1, TM, NC>(c, w - 1, G, Dn, rr, nxt, dir);
        } else if (TM == 1 || T == 1) dmv_bw_span_p<SR, -1, 1, TM, X, NC>(c, w, G, D, live, rr, x, cur, nxt, more, w - 1, Dn, dir);
        else if (TM == 2 || T == 2) dmv_bw_span_p<SR, -1, 2, TM, X, NC>(c, w, G, D, live, rr, x, cur, nxt, more, w - 1, Dn, dir);
        else dmv_bw_span_p<SR, -1, 3, TM, X, NC>(c, w, G, D, live, rr, x, cur, nxt, more, w - 1, Dn, dir);
        x.sync();
    };
    for (int w = w1 - 1; w >= w0; w -= 2) {
        one_width(w, va, vb);
        if (w - 1 >= w0) one_width(w - 1, vb, va);
    }
}

template <int SR, int DIR, int LG, int LONGSPAN, typename X>
VLG_HD void dmv_bw_width(const DmvCtx& c, int w, int t, int nd, X& x, int dir_rt = 0) {
    constexpr int G = 1 << LG;
    constexpr bool NC = LONGSPAN != 1 && X::kChartsInLds;
    const int per = nd >> LG, rr = t & (G - 1), slot = t >> LG, spans = c.Ne - w;
    const int T = (w + G - 1) >> LG;
    for (int base = 0; base < spans; base += per) {
        const bool live = base + slot < spans;
        const int i = live ? base + slot : 0;
        if (X::kSkipDeadWaves && (base + ((t & ~63) >> LG)) >= spans) continue;
        const int D = VLG_MUL24(i, c.P + 1);
        if constexpr (LONGSPAN == kSpansShort) {
            constexpr int TM = short_tmax(LG, kShortN, VLG_DP_LANES_BW);
            if (TM == 1 || T == 1) dmv_bw_span<SR, DIR, 1, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
            else if (TM == 2 || T == 2) dmv_bw_span<SR, DIR, 2, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
            else dmv_bw_span<SR, DIR, 3, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
        } else if (T == 1) dmv_bw_span<SR, DIR, 1, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
        else if (T == 2) dmv_bw_span<SR, DIR, 2, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
        else if (T == 3) dmv_bw_span<SR, DIR, 3, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
        else if (T == 4) dmv_bw_span<SR, DIR, 4, X, false, NC>(c, w, G, D, live, rr, x, 1, dir_rt);
        else dmv_bw_span<SR, DIR, 4, X, true>(c, w, G, D, live, rr, x, (T + 3) >> 2, dir_rt);   // chunks of four split points per lane
    }
}

// ---- Outside pass, long-sentence placements (value charts in the workspace): ONE stream of value reads over the whole segment ------
// With the value charts in global memory every chunk of four split points per lane began with a round trip to the L2 (~1-2 us at one
// workgroup per CU): 314 us of outside pass at N = 81 against 185 us of inside pass (3.9 us per width where N = 41 takes 1.0).  What those
// reads fetch is the inside pass's tape -- nothing writes it now -- so they need not wait for anything: the items (width, span block,
// chunk) of a segment form one sequence, and while item k is processed the value reads of item k + 1 -- the next chunk of the span, or
// the first chunk and the own cells of the next width's span -- are in flight (issued behind item k's adjoint reads, ahead of its
// stores).  Behind a width's barrier only the adjoint reads (LDS) are requested.  Same operations per result as dmv_bw_span's chunked
// form: bit-identical counts.
struct BwOwn { float2 oc, sv; float Sv, su; };
struct BwChunk { float2 vv[4]; float uu[4], xa[4], xb[4]; };

// element `word` (an index in 4-byte words, non-negative, < 2^30) of a chart in the workspace, addressed as base + unsigned 32-bit BYTE
// offset: one load with the (uniform) base in scalar registers, where base[int] costs a sign extension and a 64-bit shift-add per
// address -- 35 of the 311 vector instructions of an item of the streamed outside pass (N = 81 launch 492 -> 487 us, same bits)
template <typename T>
VLG_HD T chart_ld(const void* base, int word) {
    return *reinterpret_cast<const T*>(static_cast<const char*>(base) + static_cast<size_t>(static_cast<unsigned>(word) << 2));
}

template <typename X>
VLG_HD void dmv_bw_item_loads(const DmvCtx& c, int w, int G, int D, int r0, int DIR, bool first, BwOwn& own, BwChunk& ch) {
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    const int eA = 2 * (D + 1) + (DIR == 0 ? 1 : 0), eB = 2 * (DW + 1) + (DIR == 0 ? 0 : 1);
    const int eU = DIR == 0 ? D : D + P + w + 1, eV = DIR == 0 ? DW : D + 2;
    if (first) {
        const int kO = DIR == 0 ? DW : D + w + 1, kS = DIR == 0 ? DW : D + w, selfr = DIR == 0 ? 0 : w - 1;
        own.oc = chart_ld<float2>(c.C, 2 * kO);
        own.Sv = chart_ld<float>(c.S, kS);
        own.su = chart_ld<float>(c.C, 2 * (eU + VLG_MUL24(selfr, P)) + 1);
        own.sv = chart_ld<float2>(c.I, 2 * (eV + selfr));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int r = r0 + u * G, rc = r < w ? r : w - 1, rP = VLG_MUL24(rc, P);
        ch.uu[u] = chart_ld<float>(c.C, 2 * (eU + rP) + 1);
        ch.vv[u] = chart_ld<float2>(c.I, 2 * (eV + rc));
        ch.xa[u] = chart_ld<float>(c.C, eA + 2 * rc);
        ch.xb[u] = chart_ld<float>(c.C, eB + 2 * rc);
    }
}

template <int SR, int LG, typename X>
VLG_HD void dmv_bw_segment_s(const DmvCtx& c, int w0, int w1, int tid, int nt, X& x) {
    constexpr int G = 1 << LG;
    if (w1 <= w0) return;
    const int nd = nt >> 1;
    const bool right = x.uniform(tid >= nd);
    const int DIR = right ? 1 : 0;
    const int t = right ? tid - nd : tid;
    const int per = nd >> LG, rr = t & (G - 1), slot = t >> LG, P = c.P;
    float* gCif = reinterpret_cast<float*>(c.gCi);
    auto span_D = [&](int w_, int base_) { return VLG_MUL24(base_ + slot < c.Ne - w_ ? base_ + slot : 0, P + 1); };
    BwOwn own, own_n;
    BwChunk cur, nxt;
    int w = w1 - 1, base = 0, ch = 0;
    dmv_bw_item_loads<X>(c, w, G, span_D(w, 0), rr, DIR, true, own, cur);
    // per-span state, set at the span's first chunk
    float2 gc = make_float2(0.f, 0.f), gi = make_float2(0.f, 0.f), oc = make_float2(0.f, 0.f);
    float gs = 0.f, Sv = 0.f;
    int b0 = 0, b1 = 0, bs = 0;
    for (;;) {
        const int spans = c.Ne - w, nch = (((w + G - 1) >> LG) + 3) >> 2;
        const bool live = base + slot < spans;
        const int D = span_D(w, base), DW = D + VLG_MUL24(w, P);
        const int eA = 2 * (D + 1) + (DIR == 0 ? 1 : 0), eB = 2 * (DW + 1) + (DIR == 0 ? 0 : 1);
        const int eU = DIR == 0 ? D : D + P + w + 1, eV = DIR == 0 ? DW : D + 2;
        const int kO = DIR == 0 ? DW : D + w + 1, kS = DIR == 0 ? DW : D + w, selfr = DIR == 0 ? 0 : w - 1;
        // the item after this one (uniform over the workgroup)
        int wn = w, bn = base, chn = ch + 1;
        bool more = true;
        if (chn == nch) {
            chn = 0;
            bn = base + per;
            if (bn >= spans) { bn = 0; wn = w - 1; more = wn >= w0; }
        }
        const bool width_ends = wn != w;
        const int r0 = rr + ch * (4 * G);
        // ---- adjoint reads (LDS): the own cells at a span's first chunk, the read-modify-write targets of this chunk ----
        float ga = 0.f;
        float2 gb = make_float2(0.f, 0.f), gi_old = make_float2(0.f, 0.f);
        if (ch == 0) {
            ga = c.gCc[kO];
            gb = c.gCi[kO];
            gi_old = c.gI[kO];
            if (SR == VLG_SR_MAX) { b0 = c.bpC[kO * 2]; b1 = c.bpC[kO * 2 + 1]; bs = c.bpS[kS]; }
        }
        float o_c[4], o_ga[4], o_gb[4];
        float2 o_gi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * G, rc = r < w ? r : w - 1, rP = VLG_MUL24(rc, P);
            o_gi[u] = c.gI[eV + rc];
            o_c[u] = c.gCc[eU + rP];
            o_ga[u] = gCif[eA + 2 * rc];
            o_gb[u] = gCif[eB + 2 * rc];
        }
        // ---- the next item's value reads (workspace): in flight while this item is processed ----
        if (more) dmv_bw_item_loads<X>(c, wn, G, span_D(wn, bn), rr + chn * (4 * G), DIR, chn == 0, own_n, nxt);
        if (ch == 0) {   // span state: total adjoint of the own complete cell, the same-width term's share, gs
            oc = own.oc;
            Sv = own.Sv;
            gc = make_float2(0.f, 0.f);
            if (live && !(DIR == 1 && D == 0 && w != c.len)) gc = make_float2(gb.x, ga + gb.y);   // masked root cell: dmv.py:63
            const float self0 = adj_w<SR>(gc.x, own.su + own.sv.x, oc.x, selfr, b0);
            const float self1 = adj_w<SR>(gc.y, own.su + own.sv.y, oc.y, selfr, b1);
            gi = make_float2(gi_old.x + self0, gi_old.y + self1);   // complete adjoint of IL(j,i) | IR(i,j)
            gs = gi.x + gi.y;
        }
        float q0[4], q1[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * G;
            const bool ok = r < w;
            q0[u] = ok ? adj_w<SR>(gc.x, cur.uu[u] + cur.vv[u].x, oc.x, r, b0) : 0.f;
            q1[u] = ok ? adj_w<SR>(gc.y, cur.uu[u] + cur.vv[u].y, oc.y, r, b1) : 0.f;
        }
        x.lockstep();   // the lanes of a group sit in one wavefront: every load above precedes every store below
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = r0 + u * G;
            if (live && r < w) {
                const float ws = adj_w<SR>(gs, cur.xa[u] + cur.xb[u], Sv, r, bs);
                if (r != selfr) c.gI[eV + r] = make_float2(o_gi[u].x + q0[u], o_gi[u].y + q1[u]);
                c.gCc[eU + VLG_MUL24(r, P)] = o_c[u] + (q0[u] + q1[u]);
                gCif[eA + 2 * r] = o_ga[u] + ws;
                gCif[eB + 2 * r] = o_gb[u] + ws;
            }
        }
        if (chn == 0 && live && rr == 0) c.gI[kO] = gi;   // the span's last chunk: == d logZ / d attach[j,i,:] | attach[i,j,:]
        if (width_ends) x.sync();
        if (!more) break;
        if (chn == 0) { own.oc = own_n.oc; own.sv = own_n.sv; own.Sv = own_n.Sv; own.su = own_n.su; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { cur.vv[u] = nxt.vv[u]; cur.uu[u] = nxt.uu[u]; cur.xa[u] = nxt.xa[u]; cur.xb[u] = nxt.xb[u]; }
        w = wn; base = bn; ch = chn;
    }
}

// widths w1-1 ... w0 of one segment, descending, one barrier per width
template <int SR, int LG, int LONGSPAN, typename X>
VLG_HD void dmv_bw_segment(const DmvCtx& c, int w0, int w1, int tid, int nt, X& x) {
    const int nd = nt >> 1;
    const bool right = x.uniform(tid >= nd);
    const int t = right ? tid - nd : tid;
    for (int w = w1 - 1; w >= w0; --w) {
#ifndef VLG_DIR_TEMPLATE_BW
        // Round 4: the direction is a wave-uniform RUN-TIME value in the outside pass -- both halves of the workgroup execute one copy
        // of the code, so each instruction-cache line serves eight wavefronts instead of four (the kernel is ~200 KB and 11 % of its wave
        // cycles wait for instructions); the direction-dependent indices become a few scalar selects per span.  Fused launch 75.8 ->
        // 75.5 us, bit-identical, half the outside-pass code.  (The same change in the inside pass costs 6 us: its direction-dependent
        // predicates sit per term.)
        dmv_bw_width<SR, -1, LG, LONGSPAN>(c, w, t, nd, x, right ? 1 : 0);
#else
        if (right) dmv_bw_width<SR, 1, LG, LONGSPAN>(c, w, t, nd, x);
        else dmv_bw_width<SR, 0, LG, LONGSPAN>(c, w, t, nd, x);
#endif
        x.sync();
    }
}

// (Round 4, measured and not kept: the outside pass has no cross-lane reduction, so its lane groups need not be powers of two --
//  G = 7 / 10 / 12 between them cut the split points per lane from 3 to 2 for the widths 17 ... 24 and from 2 to 1 for 5 ... 7 at
//  N = 41, with bit-identical results.  Three more segment instantiations made the launch 1 us SLOWER: each width already runs
//  code the instruction cache has not seen since the previous launch, and ten segments x two directions x four unrollings is more
//  of it.  Also measured: the read-modify-writes of the adjoint charts as non-returning ds_add_f32 (one writer per word and phase,
//  so the same sums): 77 -> 200 us, LDS float atomics are that slow; the four adjoint loads per split point replaced by register
//  copies (wrong results, an upper bound for interleaving value and adjoint cells): 77.3 -> 74.4 us.)
template <int SR, int LONGSPAN = true, typename X>
VLG_HD void dmv_bw_all(const DmvCtx& c, int tid, int nt, X& x) {
    const Sched sc = make_sched(c.Ne, nt >> 1, VLG_DP_LANES_BW);
#ifndef VLG_NO_BW_PREFETCH
    if constexpr (LONGSPAN == kSpansShort) {   // the short-sentence image: value reads one width ahead (dmv_bw_segment_p)
        dmv_bw_segment_p<SR, 6>(c, sc.first[6], sc.first[7], tid, nt, x);
        dmv_bw_segment_p<SR, 5>(c, sc.first[5], sc.first[6], tid, nt, x);
        dmv_bw_segment_p<SR, 4>(c, sc.first[4], sc.first[5], tid, nt, x);
        dmv_bw_segment_p<SR, 3>(c, sc.first[3], sc.first[4], tid, nt, x);
        dmv_bw_segment_p<SR, 2>(c, sc.first[2], sc.first[3], tid, nt, x);
        dmv_bw_segment_p<SR, 1>(c, sc.first[1], sc.first[2], tid, nt, x);
        dmv_bw_segment_p<SR, 0>(c, sc.first[0], sc.first[1], tid, nt, x);
        return;
    }
#endif
#ifndef VLG_NO_BW_STREAM
    if constexpr (LONGSPAN == kSpansLong) {   // long-sentence placements: the value reads as one stream over each segment (dmv_bw_segment_s)
        dmv_bw_segment_s<SR, 6>(c, sc.first[6], sc.first[7], tid, nt, x);
        dmv_bw_segment_s<SR, 5>(c, sc.first[5], sc.first[6], tid, nt, x);
        dmv_bw_segment_s<SR, 4>(c, sc.first[4], sc.first[5], tid, nt, x);
        dmv_bw_segment_s<SR, 3>(c, sc.first[3], sc.first[4], tid, nt, x);
        dmv_bw_segment_s<SR, 2>(c, sc.first[2], sc.first[3], tid, nt, x);
        dmv_bw_segment_s<SR, 1>(c, sc.first[1], sc.first[2], tid, nt, x);
        dmv_bw_segment_s<SR, 0>(c, sc.first[0], sc.first[1], tid, nt, x);
        return;
    }
#endif
    dmv_bw_segment<SR, 6, LONGSPAN>(c, sc.first[6], sc.first[7], tid, nt, x);
    dmv_bw_segment<SR, 5, LONGSPAN>(c, sc.first[5], sc.first[6], tid, nt, x);
    dmv_bw_segment<SR, 4, LONGSPAN>(c, sc.first[4], sc.first[5], tid, nt, x);
    dmv_bw_segment<SR, 3, LONGSPAN>(c, sc.first[3], sc.first[4], tid, nt, x);
    dmv_bw_segment<SR, 2, LONGSPAN>(c, sc.first[2], sc.first[3], tid, nt, x);
    dmv_bw_segment<SR, 1, LONGSPAN>(c, sc.first[1], sc.first[2], tid, nt, x);
    dmv_bw_segment<SR, 0, LONGSPAN>(c, sc.first[0], sc.first[1], tid, nt, x);
}

// ================================================================================================
// DepTree (plain first-order Eisner, single root), deptree.py:25-76.  Cells are single floats;
//   T(i,j) = (+)_r CR(i,i+r) + CL(j,i+r+1)  (stored at S[i*P+j]);  IL(j,i) = T + arc[j,i]; IR(i,j) = T + arc[i,j]
//   CL(j,i) = (+)_r CL(i+r,i) + IL(j,i+r) ;  CR(i,j) = (+)_r IR(i,i+1+r) + CR(i+1+r,j) ;  CR(0,w)=zero unless w==len
// ================================================================================================
struct DepCtx {
    int Ne, len, P;
    float* C;
    float* I;       // pre-loaded with arc scores
    float* S;
    float* gCc;
    float* gCi;
    float* gI;
    unsigned char* bpS;
    unsigned char* bpC;
};

// Inside, one span, one direction (same organisation as dmv_fw_span): DIR 0 = T -> IL(j,i) -> CL(j,i), DIR 1 = T -> IR(i,j) ->
// CR(i,j).  Both directions need T(i,j); each computes it (same lanes, same butterfly tree: identical bits), DIR 0 stores it.
template <int SR, bool BWD, int DIR, int TU, typename X>
VLG_HD void dep_fw_span(const DepCtx& c, int w, int G, int D, bool live, int rr, X& x) {
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    const int eA = D + 1, eB = DW + 1;                        // CR(i, i+r), CL(j, i+r+1)
    const int eU = DIR == 0 ? D : D + P + w + 1;              // CL(i+r, i) | CR(i+1+r, j)   (stride P)
    const int eV = DIR == 0 ? DW : D + 2;                     // IL(j, i+r) | IR(i, i+1+r)
    const int kO = DIR == 0 ? DW : D + w + 1;                 // own slot: IL(j,i) / CL(j,i) | IR(i,j) / CR(i,j)
    const float aX = c.I[kO];                                 // arc score, staged at load
    const float cX = DIR == 0 ? c.C[D] : c.C[DW + w + 1];    // CL(i,i) | CR(j,j)
    float m[2], s[2] = {0.f, 0.f};
    int am[2];
    if (TU > 0) {
        float t[TU > 0 ? TU : 1][2];
#pragma unroll
        for (int u = 0; u < TU; ++u) {
            const int r = rr + u * G, rc = r < w ? r : w - 1;
            const float a = c.C[eA + rc], b = c.C[eB + rc], uu = c.C[eU + VLG_MUL24(rc, P)], vv = c.I[eV + rc];
            const bool v0 = r < w, v1 = DIR == 0 ? (v0 && r >= 1) : (r <= w - 2);
            t[u][0] = v0 ? a + b : VLG_LOWEST;
            t[u][1] = v1 ? uu + vv : VLG_LOWEST;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            m[k] = t[0][k];
            am[k] = rr;
            if (SR == VLG_SR_LOG) {
#pragma unroll
                for (int u = 1; u < TU; ++u) m[k] = fmaxf(m[k], t[u][k]);
            } else {
#pragma unroll
                for (int u = 1; u < TU; ++u)
                    if (t[u][k] > m[k]) { m[k] = t[u][k]; am[k] = rr + u * G; }
            }
        }
        if (SR == VLG_SR_MAX) x.template allreduce_argmax<2>(m, am, G);
        else x.template allreduce_max<2>(m, G);
        if (SR == VLG_SR_LOG) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int u = 0; u < TU; ++u) s[k] += VLG_EXP(t[u][k] - m[k]);
            x.template allreduce_sum<2>(s, G);
        }
    } else {
        // (left as a loop over single split points: the chunked form of dmv_fw_span costs this kernel registers -- and with them
        //  resident workgroups at large batches: B = 4096 1.27 ms vs 0.78 ms -- for nothing at N = 81, where its charts sit in LDS)
        for (int k = 0; k < 2; ++k) { m[k] = VLG_LOWEST; am[k] = 0; }
        for (int r = rr; r < w; r += G) {
            upd_max(m[0], am[0], c.C[eA + r] + c.C[eB + r], r);
            if (DIR == 0 ? r >= 1 : r <= w - 2) upd_max(m[1], am[1], c.C[eU + VLG_MUL24(r, P)] + c.I[eV + r], r);
        }
        if (SR == VLG_SR_MAX) x.template allreduce_argmax<2>(m, am, G);
        else x.template allreduce_max<2>(m, G);
        if (SR == VLG_SR_LOG) {
            for (int r = rr; r < w; r += G) {
                s[0] += VLG_EXP(c.C[eA + r] + c.C[eB + r] - m[0]);
                if (DIR == 0 ? r >= 1 : r <= w - 2) s[1] += VLG_EXP(c.C[eU + VLG_MUL24(r, P)] + c.I[eV + r] - m[1]);
            }
            x.template allreduce_sum<2>(s, G);
        }
    }
    const float T = SR == VLG_SR_LOG ? m[0] + VLG_LOG(s[0]) : m[0];
    const float In = aX + T;   // deptree.py:58,62
    int b0;
    float Cv = fold_term<SR>(m[1], s[1], am[1], cX + In, DIR == 0 ? 0 : w - 1, DIR == 0, b0);
    if (DIR == 1 && D == 0 && w != c.len) Cv = VLG_NEGINF;   // deptree.py:71-72
    if (live && rr == 0) {
        c.I[kO] = In;
        c.C[kO] = Cv;
        if (BWD) {
            if (DIR == 0 && SR != VLG_SR_MAX) c.S[D + w] = T;   // T(i,j) at [i][j]: the outside replay's tape; the Max semiring walks back-pointers
            if (SR == VLG_SR_MAX) {
                if (DIR == 0) c.bpS[D + w] = (unsigned char)am[0];
                c.bpC[kO] = (unsigned char)b0;
            }
        }
    }
}

template <int SR, bool BWD, int DIR, int LG, int SPANS, typename X>
VLG_HD void dep_fw_width(const DepCtx& c, int w, int t, int nd, X& x) {
    constexpr int G = 1 << LG;
    const int per = nd >> LG, rr = t & (G - 1), slot = t >> LG, spans = c.Ne - w;
    const int T = (w + G - 1) >> LG;
    for (int base = 0; base < spans; base += per) {
        const bool live = base + slot < spans;
        const int i = live ? base + slot : 0;
        if (X::kSkipDeadWaves && (base + ((t & ~63) >> LG)) >= spans) continue;
        const int D = VLG_MUL24(i, c.P + 1);
        if constexpr (SPANS == kSpansShort) {
            constexpr int TM = short_tmax(LG, kShortN, VLG_DP_LANES_FW);
            if (TM == 1 || T == 1) dep_fw_span<SR, BWD, DIR, 1>(c, w, G, D, live, rr, x);
            else if (TM == 2 || T == 2) dep_fw_span<SR, BWD, DIR, 2>(c, w, G, D, live, rr, x);
            else dep_fw_span<SR, BWD, DIR, 3>(c, w, G, D, live, rr, x);
        } else if (T == 1) dep_fw_span<SR, BWD, DIR, 1>(c, w, G, D, live, rr, x);
        else if (T == 2) dep_fw_span<SR, BWD, DIR, 2>(c, w, G, D, live, rr, x);
        else if (T == 3) dep_fw_span<SR, BWD, DIR, 3>(c, w, G, D, live, rr, x);
        else if (T == 4) dep_fw_span<SR, BWD, DIR, 4>(c, w, G, D, live, rr, x);
        else dep_fw_span<SR, BWD, DIR, 0>(c, w, G, D, live, rr, x);
    }
}

template <int SR, bool BWD, int LG, int SPANS, typename X>
VLG_HD void dep_fw_segment(const DepCtx& c, int w0, int w1, int tid, int nt, X& x) {
    const int nd = nt >> 1;
    const bool right = x.uniform(tid >= nd);
    const int t = right ? tid - nd : tid;
    for (int w = w0; w < w1; ++w) {
        if (right) dep_fw_width<SR, BWD, 1, LG, SPANS>(c, w, t, nd, x);
        else dep_fw_width<SR, BWD, 0, LG, SPANS>(c, w, t, nd, x);
        x.sync();
    }
}

template <int SR, bool BWD, int SPANS, typename X>
VLG_HD void dep_fw_all(const DepCtx& c, int tid, int nt, X& x) {
    const Sched sc = make_sched(c.Ne, nt >> 1, VLG_DP_LANES_FW);
    dep_fw_segment<SR, BWD, 0, SPANS>(c, sc.first[0], sc.first[1], tid, nt, x);
    dep_fw_segment<SR, BWD, 1, SPANS>(c, sc.first[1], sc.first[2], tid, nt, x);
    dep_fw_segment<SR, BWD, 2, SPANS>(c, sc.first[2], sc.first[3], tid, nt, x);
    dep_fw_segment<SR, BWD, 3, SPANS>(c, sc.first[3], sc.first[4], tid, nt, x);
    dep_fw_segment<SR, BWD, 4, SPANS>(c, sc.first[4], sc.first[5], tid, nt, x);
    dep_fw_segment<SR, BWD, 5, SPANS>(c, sc.first[5], sc.first[6], tid, nt, x);
    dep_fw_segment<SR, BWD, 6, SPANS>(c, sc.first[6], sc.first[7], tid, nt, x);
}

// The self term of IL(j,i) / IR(i,j): the weight of the same-width term in CL(j,i) / CR(i,j) (r = 0 | w-1).  gI never
// receives it during the sweep -- T(i,j) feeds both and the two directions of a span run concurrently, so neither may
// write what the other reads; each recomputes both self terms, and dep_run's output stage adds them once at the end.
template <int SR>
VLG_HD float dep_self(const DepCtx& c, int D, int w, int dir) {
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    const int kO = dir == 0 ? DW : D + w + 1;
    float g = c.gCc[kO] + c.gCi[kO];
    if (dir == 1 && D == 0 && w != c.len) g = 0.f;   // masked root cell (deptree.py:71-72)
    const float cX = dir == 0 ? c.C[D] : c.C[DW + w + 1];
    const int bp = SR == VLG_SR_MAX ? c.bpC[kO] : 0;
    return adj_w<SR>(g, cX + c.I[kO], c.C[kO], dir == 0 ? 0 : w - 1, bp);
}

// Outside, one span, one direction: DIR 0 scatters CL(j,i)'s adjoint (-> IL(j,i+r), CL(i+r,i)) and T's into the CR(i,.)
// operands; DIR 1 scatters CR(i,j)'s (-> IR(i,i+1+r), CR(i+1+r,j)) and T's into the CL(j,.) operands.
template <int SR, int DIR, int TU, typename X>
VLG_HD void dep_bw_span(const DepCtx& c, int w, int G, int D, bool live, int rr, X& x, int nchunk = 1) {
    const int P = c.P, DW = D + VLG_MUL24(w, P);
    const int eU = DIR == 0 ? D : D + P + w + 1;      // CL(i+r, i) | CR(i+1+r, j)   (stride P)
    const int eV = DIR == 0 ? DW : D + 2;             // IL(j, i+r) | IR(i, i+1+r)
    const int eX = DIR == 0 ? D + 1 : DW + 1;         // this direction's T operand: CR(i, i+r) | CL(j, i+r+1)
    const int kO = DIR == 0 ? DW : D + w + 1;
    const int selfr = DIR == 0 ? 0 : w - 1;
    float g = c.gCc[kO] + c.gCi[kO];
    if (!live || (DIR == 1 && D == 0 && w != c.len)) g = 0.f;
    const float oc = c.C[kO], Tv = c.S[D + w];
    const float gs = live ? c.gI[DW] + c.gI[D + w + 1] + dep_self<SR>(c, D, w, 0) + dep_self<SR>(c, D, w, 1) : 0.f;
    if (TU > 0) {
        // nchunk > 1: chunks of TU split points per lane, each with its loads ahead of its stores (see dmv_bw_span)
        for (int ch = 0; ch < nchunk; ++ch) {
            const int r0 = rr + ch * (TU * G);
            float uu[TU > 0 ? TU : 1], vv[TU > 0 ? TU : 1], xa[TU > 0 ? TU : 1], xb[TU > 0 ? TU : 1];
            float o_gi[TU > 0 ? TU : 1], o_c[TU > 0 ? TU : 1], o_g[TU > 0 ? TU : 1];
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int r = r0 + u * G, rc = r < w ? r : w - 1;
                uu[u] = c.C[eU + VLG_MUL24(rc, P)];
                vv[u] = c.I[eV + rc];
                xa[u] = c.C[D + 1 + rc];
                xb[u] = c.C[DW + 1 + rc];
                o_gi[u] = c.gI[eV + rc];
                o_c[u] = c.gCc[eU + VLG_MUL24(rc, P)];
                o_g[u] = c.gCi[eX + rc];
            }
            x.lockstep();   // the lanes of a group sit in one wavefront: every load above precedes every store below
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int r = r0 + u * G;
                if (live && r < w) {
                    const float wc = adj_w<SR>(g, uu[u] + vv[u], oc, r, 0);
                    const float wt = adj_w<SR>(gs, xa[u] + xb[u], Tv, r, 0);
                    if (r != selfr) c.gI[eV + r] = o_gi[u] + wc;
                    c.gCc[eU + VLG_MUL24(r, P)] = o_c[u] + wc;
                    c.gCi[eX + r] = o_g[u] + wt;
                }
            }
        }
        return;
    }
    for (int r = rr; r < w; r += G) {
        if (!live) break;
        const float wc = adj_w<SR>(g, c.C[eU + VLG_MUL24(r, P)] + c.I[eV + r], oc, r, 0);
        const float wt = adj_w<SR>(gs, c.C[D + 1 + r] + c.C[DW + 1 + r], Tv, r, 0);
        if (r != selfr) c.gI[eV + r] += wc;
        c.gCc[eU + VLG_MUL24(r, P)] += wc;
        c.gCi[eX + r] += wt;
    }
}

template <int SR, int DIR, int LG, int SPANS, typename X>
VLG_HD void dep_bw_width(const DepCtx& c, int w, int t, int nd, X& x) {
    constexpr int G = 1 << LG;
    const int per = nd >> LG, rr = t & (G - 1), slot = t >> LG, spans = c.Ne - w;
    const int T = (w + G - 1) >> LG;
    for (int base = 0; base < spans; base += per) {
        const bool live = base + slot < spans;
        const int i = live ? base + slot : 0;
        if (X::kSkipDeadWaves && (base + ((t & ~63) >> LG)) >= spans) continue;
        const int D = VLG_MUL24(i, c.P + 1);
        if constexpr (SPANS == kSpansShort) {
            constexpr int TM = short_tmax(LG, kShortN, VLG_DP_LANES_BW);
            if (TM == 1 || T == 1) dep_bw_span<SR, DIR, 1>(c, w, G, D, live, rr, x);
            else if (TM == 2 || T == 2) dep_bw_span<SR, DIR, 2>(c, w, G, D, live, rr, x);
            else dep_bw_span<SR, DIR, 3>(c, w, G, D, live, rr, x);
        } else if (T == 1) dep_bw_span<SR, DIR, 1>(c, w, G, D, live, rr, x);
        else if (T == 2) dep_bw_span<SR, DIR, 2>(c, w, G, D, live, rr, x);
        else if (T == 3) dep_bw_span<SR, DIR, 3>(c, w, G, D, live, rr, x);
        else if (T == 4) dep_bw_span<SR, DIR, 4>(c, w, G, D, live, rr, x);
        else dep_bw_span<SR, DIR, 4>(c, w, G, D, live, rr, x, (T + 3) >> 2);
    }
}

template <int SR, int LG, int SPANS, typename X>
VLG_HD void dep_bw_segment(const DepCtx& c, int w0, int w1, int tid, int nt, X& x) {
    const int nd = nt >> 1;
    const bool right = x.uniform(tid >= nd);
    const int t = right ? tid - nd : tid;
    for (int w = w1 - 1; w >= w0; --w) {
        if (right) dep_bw_width<SR, 1, LG, SPANS>(c, w, t, nd, x);
        else dep_bw_width<SR, 0, LG, SPANS>(c, w, t, nd, x);
        x.sync();
    }
}

template <int SR, int SPANS, typename X>
VLG_HD void dep_bw_all(const DepCtx& c, int tid, int nt, X& x) {
    const Sched sc = make_sched(c.Ne, nt >> 1, VLG_DP_LANES_BW);
    dep_bw_segment<SR, 6, SPANS>(c, sc.first[6], sc.first[7], tid, nt, x);
    dep_bw_segment<SR, 5, SPANS>(c, sc.first[5], sc.first[6], tid, nt, x);
    dep_bw_segment<SR, 4, SPANS>(c, sc.first[4], sc.first[5], tid, nt, x);
    dep_bw_segment<SR, 3, SPANS>(c, sc.first[3], sc.first[4], tid, nt, x);
    dep_bw_segment<SR, 2, SPANS>(c, sc.first[2], sc.first[3], tid, nt, x);
    dep_bw_segment<SR, 1, SPANS>(c, sc.first[1], sc.first[2], tid, nt, x);
    dep_bw_segment<SR, 0, SPANS>(c, sc.first[0], sc.first[1], tid, nt, x);
}

// ================================================================================================
// I/O policies of the DMV driver: how one sentence's potentials enter the load stage and where its
// expected counts go.  `h`, `ch` are positions in the root-augmented sentence (0 = root).
// ================================================================================================
// (a) root-merged potentials -- what `DMV1o([dec, attach], lengths)` takes (distributions.py:245-251).
template <typename In>
struct MergedIO {
    const typename In::T* dec;      // [N][2][2][2]
    const typename In::T* attach;   // [N][N][2]
    int N;
    float* gdec;                    // may be null (decode mode)
    float* gatt;
    long long* heads;               // may be null
    VLG_HDM int out_extent(int) const { return N; }   // outputs cover the padded square: zeros beyond the sentence
    VLG_HDM float ld_dec(int i) const { return pot_clamp(In::ld(dec, i)); }
    VLG_HDM float2 ld_attach(int h, int ch) const { return pot_clamp2(In::ld2(attach, ((size_t)h * N + ch) * 2)); }
    VLG_HDM void st_attach(int h, int ch, float2 g) const {
        if (gatt) *reinterpret_cast<float2*>(gatt + ((size_t)h * N + ch) * 2) = g;
        if (heads && g.x + g.y != 0.f) heads[ch] = h;
    }
    VLG_HDM bool wants_dec() const { return gdec != nullptr; }
    VLG_HDM void st_dec(int h, int k, float g) const { gdec[h * 8 + k] = g; }
    VLG_HDM void clear_heads(int tid, int nt) const {
        if (heads) for (int i = tid; i < N; i += nt) heads[i] = 0;
    }
};

// (b) the scorer's rule tables -- SURVEY.md section 8(f)1.  Folds into the load stage what the reference does
// with five tensor ops before the DP (src/model/ldndmv.py:189-209): gather attach_rule by the child's token id,
// pick the LEFT / RIGHT slice by tril / triu masks, mask function-word heads, gather root by token id, and
// DMV1o.merge.  Expected counts go back in RULE space (the adjoint of that gather is a scatter-add over
// repeated tokens: fp32 atomics into this sentence's slice; the caller zero-fills it).
template <typename In>
struct RuleIO {
    const typename In::T* rule;     // attach_rule of this sentence [L][T][2(dir)][2(val)]
    const typename In::T* dec;      // [L][2][2][2]
    const typename In::T* root;     // root log-probs [T]
    const long long* token;         // [L]
    const unsigned char* head_mask; // [L] non-zero = this word takes no children (function_mask); may be null
    int L, T;
    float fill;                     // the reference's -INF fill for masked heads (src/__init__.py:110)
    float* g_rule;                  // [L][T][2][2], zero-filled by the caller; may be null
    float* g_dec;                   // [L][2][2][2]
    float* g_root;                  // [T], zero-filled by the caller
    long long* heads;               // [L+1]; may be null
    VLG_HDM int out_extent(int Ne) const { return Ne; }   // only real positions: everything else is already zero
    VLG_HDM float ld_dec(int i) const {
        const int h = i >> 3, k = i & 7;
        if (h == 0) return (k >> 2) == 1 ? 0.f : VLG_NEGINF;                 // dec_wroot[0,RIGHT] = one, else zero
        return pot_clamp(In::ld(dec, (h - 1) * 8 + k));
    }
    VLG_HDM size_t rule_index(int h, int ch) const {                          // h, ch >= 1, h != ch
        return ((((size_t)(h - 1) * T + (size_t)token[ch - 1]) * 2 + (ch < h ? 0 : 1)) * 2);
    }
    VLG_HDM float2 ld_attach(int h, int ch) const {
        if (ch == 0) return make_float2(VLG_NEGINF, VLG_NEGINF);             // nobody attaches the root
        if (h == 0) return make_float2(VLG_NEGINF, pot_clamp(In::ld(root, (size_t)token[ch - 1])));   // [0, c, NOCHILD] = root
        if (head_mask && head_mask[h - 1]) return make_float2(fill, fill);
        return pot_clamp2(In::ld2(rule, rule_index(h, ch)));
    }
    VLG_HDM void st_attach(int h, int ch, float2 g) const {
        if (heads && g.x + g.y != 0.f) heads[ch] = h;
        if (!g_rule || ch == 0 || h == ch) return;
        if (h == 0) { if (g.y != 0.f) VLG_ATOMIC_ADD(g_root + token[ch - 1], g.y); return; }
        if (head_mask && head_mask[h - 1]) return;                           // masked_fill_: no gradient
        float* p = g_rule + rule_index(h, ch);
        if (g.x != 0.f) VLG_ATOMIC_ADD(p, g.x);
        if (g.y != 0.f) VLG_ATOMIC_ADD(p + 1, g.y);
    }
    VLG_HDM bool wants_dec() const { return g_dec != nullptr; }
    VLG_HDM void st_dec(int h, int k, float g) const {
        if (h >= 1) g_dec[(h - 1) * 8 + k] = g;
    }
    VLG_HDM void clear_heads(int tid, int nt) const {
        if (heads) for (int i = tid; i < L + 1; i += nt) heads[i] = 0;
    }
};

// ================================================================================================
// Whole-sentence drivers: the body of one workgroup.  `x.sync()` is __syncthreads() on the GPU and
// the token barrier of the host phase emulator in the CPU tests.  Pointers in the context are
// already carved (LDS and/or workspace); dec/attach/gdec/gatt/logZ point at THIS sentence.
// ================================================================================================
// ------------------------------------------------------------------------------------------------
// Best tree from the Max semiring's back-pointers, by walking the derivation from the root span instead of replaying
// the whole outside pass with one-hot weights (which costs as much as the inside pass; the walk visits ~4N spans).
// One lane walks; the stack lives in the (unused) gCc chart.  Every attachment it meets sets gI[...] = g at that
// (head, child, valence), which the common output loop turns into the one-hot attach tensor, the head vector and the GO counts
// (row sums of gI); every width-0 complete span it ends in is a STOP decision of that head, direction and valence (dmv.py:39-40):
// gdecs[h][dir][v] = g.  Together these are ALL the counts of the Max semiring's gradient (round 4: the one-hot replay of the
// outside pass -- as long as the inside pass, 87 us against 58 us per launch at B = 256, L = 40 -- is no longer run for it).
//   entry = kind | a << 2 | b << 10 | v << 18     kind: 0 CL(a=head, b=left end)   1 CR(a=head, b=right end)
//                                                       2 IL(a=head j, b=child)     3 IR(a=head i, b=child)
//   recurrences as in dmv_fw_span (valence index 0 = .x HASCHILD, 1 = .y NOCHILD):
//   CL(j,i).v = CL(i+r,i).NC + IL(j,i+r).v           CR(i,j).v = IR(i,i+1+r).v + CR(i+1+r,j).NC
//   IL(j,k).v = attach + CR(k,k+r).NC + CL(j,k+r+1).HC        IR(i,k).v = attach + CR(i,i+r).HC + CL(k,i+r+1).NC
// ------------------------------------------------------------------------------------------------
VLG_HD int walk_entry(int kind, int a, int b, int v) { return kind | (a << 2) | (b << 10) | (v << 18); }

VLG_HD void dmv_walk(const DmvCtx& c, float g) {
    const int P = c.P;
    int* stack = reinterpret_cast<int*>(c.gCc);   // at most one pending sibling per open span: depth <= 2N (layout: 2N + 4)
    int top = 0;
    int e = walk_entry(1, 0, c.len, 1);           // CR(0, len).NOCHILD: what logZ reads (dmv.py:65)
    // the span being expanded stays in a register and one child is followed directly; only its sibling goes through
    // the stack, so a step costs one dependent LDS read (the back-pointer), not a push-pop round trip as well
    for (;;) {
        const int kind = e & 3, a = (e >> 2) & 255, b = (e >> 10) & 255, v = (e >> 18) & 1;
        if (kind < 2 && a == b) {   // width-0 complete span: a leaf = the STOP decision of head a towards kind (0 LEFT, 1 RIGHT) at valence v
            c.gdecs[a * 4 + kind * 2 + v] = g;
            if (top == 0) break;
            e = stack[--top];
            continue;
        }
        int sib;
        if (kind == 0) {            // CL(head a, left end b)
            const int r = c.bpC[(a * P + b) * 2 + v];
            sib = walk_entry(0, b + r, b, 1);
            e = walk_entry(2, a, b + r, v);
        } else if (kind == 1) {     // CR(head a, right end b)
            const int r = c.bpC[(a * P + b + 1) * 2 + v];
            sib = walk_entry(1, a + 1 + r, b, 1);
            e = walk_entry(3, a, a + 1 + r, v);
        } else if (kind == 2) {     // IL(head a, child b): child to the left
            reinterpret_cast<float*>(c.gI + a * P + b)[v] = g;
            const int r = c.bpS[a * P + b];
            sib = walk_entry(1, b, b + r, 1);
            e = walk_entry(0, a, b + r + 1, 0);
        } else {                    // IR(head a, child b): child to the right
            reinterpret_cast<float*>(c.gI + a * P + b + 1)[v] = g;
            const int r = c.bpS[a * P + b];
            sib = walk_entry(1, a, a + r, 0);
            e = walk_entry(0, b, a + r + 1, 1);
        }
        stack[top++] = sib;
    }
}

template <int SR, bool BWD, int LONGSPAN = false, typename IO, typename X>
VLG_HD void dmv_run(const DmvCtx& c, const IO& io, float glogZ, float* logZ, int tid, int nt, X& x) {
    const int Ne = c.Ne, P = c.P, len = c.len;
    // ---- stage: charts to the semiring zero (dmv.py:34-35), potentials into fast memory -------------------------
    // The potentials' global loads are issued first and land while the charts are being filled, so the launch pays
    // one memory latency, not one per dependent step.  Incomplete-span slots are pre-loaded with attach + dec[...,GO]
    // (dmv.py:36-37); the width-0 complete spans with the STOP scores (dmv.py:39-40).  This folds the reference's
    // attach_left / attach_right temporaries into the load stage.
    const float2 zz = make_float2(VLG_NEGINF, VLG_NEGINF), oo = make_float2(0.f, 0.f);
    const int NN = Ne * Ne;
    const float inv_ne = 1.0f / (float)Ne;   // (idx + 0.5) * inv_ne truncates to idx / Ne exactly for idx < 2^16 (margin 0.5 / idx >> 2^-23)
    constexpr int KC = 4;                    // attach cells per lane in flight
    for (int base = 0; base < NN; base += KC * nt) {
        float2 areg[KC];
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int idx = base + tid + k * nt;
            const int h = (int)(((float)idx + 0.5f) * inv_ne), ch = idx - h * Ne;
            areg[k] = (idx < NN && ch != h) ? io.ld_attach(h, ch) : zz;
        }
        if (base == 0) {
            for (int i = tid; i < Ne * 8; i += nt) c.decs[i] = io.ld_dec(i) * VLG_LOG2E;
            for (int i = tid; i < Ne * P; i += nt) {
                c.C[i] = zz;
                c.I[i] = zz;
            }
            x.sync();
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int idx = base + tid + k * nt;
            if (idx >= NN) continue;
            const int h = (int)(((float)idx + 0.5f) * inv_ne), ch = idx - h * Ne;
            const float* d = c.decs + h * 8;
            if (ch == h) {
                c.C[h * P + h] = make_float2(d[1], d[3]);        // CL(h,h).v = dec[h,LEFT ,v,STOP]
                c.C[h * P + h + 1] = make_float2(d[5], d[7]);    // CR(h,h).v = dec[h,RIGHT,v,STOP]
            } else if (ch < h) {
                c.I[h * P + ch] = make_float2(fmaf(areg[k].x, VLG_LOG2E, d[0]), fmaf(areg[k].y, VLG_LOG2E, d[2]));
            } else {
                c.I[h * P + ch + 1] = make_float2(fmaf(areg[k].x, VLG_LOG2E, d[4]), fmaf(areg[k].y, VLG_LOG2E, d[6]));
            }
        }
    }
    x.sync();
    // ---- inside -----------------------------------------------------------------------------------
#if defined(VLG_STAMP) && defined(__HIPCC__)
    unsigned long long st_body = 0, st_sync = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_stage = 0, st_end = 0;
    unsigned long long st_bbody = 0, st_bsync = 0;
    {   // (per-width body / barrier split is no longer taken: the width loop lives inside the segment functions)
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        dmv_fw_all<SR, BWD, LONGSPAN>(c, tid, nt, x);
        st_body += __builtin_amdgcn_s_memtime() - a;
    }
#else
    dmv_fw_all<SR, BWD, LONGSPAN>(c, tid, nt, x);
#endif
    if (tid == 0) *logZ = c.C[len + 1].y * VLG_LN2;   // CR(0,len).NOCHILD, dmv.py:65
    if (!BWD) return;
    // ---- outside: adjoint replay ------------------------------------------------------------------
    if (c.C2 != c.C) {   // overlay placement (DmvLayout mode 1): the adjoint charts take over the LDS the value charts
        for (int i = tid; i < Ne * P; i += nt) {   // occupied -- copy those out to the workspace first (coalesced)
            c.C2[i] = c.C[i];
            c.I2[i] = c.I[i];
        }
        x.sync();
    }
    for (int i = tid; i < Ne * P; i += nt) {
        c.gI[i] = oo;
        if (SR != VLG_SR_MAX) { c.gCc[i] = 0.f; c.gCi[i] = oo; }
    }
    if (SR == VLG_SR_MAX) for (int i = tid; i < Ne * 4; i += nt) c.gdecs[i] = 0.f;
    x.sync();
    DmvCtx cb = c;   // the outside pass's view: value charts where the layout keeps them for this pass
    cb.C = c.C2;
    cb.I = c.I2;
    if (SR == VLG_SR_MAX) {   // the Max semiring's gradient IS the best tree: walk the back-pointers (compile-time: no one-hot replay is built)
#ifndef VLG_ABL_NOWALK       // (tools/ ablation: the one-lane walk is 21 of the 56 us of a Viterbi launch at B = 256, L = 40 -- ~400 cycles per
                             //  span, one dependent LDS read each.  Two parallel forms were built and measured in round 4, both 60 us: a
                             //  level-synchronous walk (one lane per open span, an LDS fetch-add for the next level's queue slots, a barrier per
                             //  level) and that walk handing over to one serial walker per wavefront once eight sub-derivations are open.
                             //  Neither finds parallelism: with the width-0 spans pruned a derivation is a chain -- most spans have ONE
                             //  non-leaf child -- so the frontier stays at 1-3 spans for ~100 levels, each dearer than a serial step.  A third form
                             //  -- still one lane, but width-0 spans handled where they are produced (half the iterations) and every span's
                             //  back-pointer read at production and carried through the stack (a pop is one LDS round trip) -- measured 57.8 us
                             //  against 55.3-56.0.  A fourth -- the walk on every lane of the first wavefront with its state pinned to scalar
                             //  registers (x.uniform on every LDS value: scalar address arithmetic and branches) -- 55.7 us: no change either.
                             //  ~300 cycles per span whichever way its instructions are issued or its reads are grouped.)
        if (tid == 0) dmv_walk(cb, glogZ);
#endif
        x.sync();
    } else {
    if (tid == 0) c.gCc[len + 1] = glogZ;
    x.sync();
#if defined(VLG_STAMP) && defined(__HIPCC__)
    {
        const unsigned long long a = __builtin_amdgcn_s_memtime();
        dmv_bw_all<SR, LONGSPAN>(cb, tid, nt, x);
        st_bbody += __builtin_amdgcn_s_memtime() - a;
    }
    st_end = __builtin_amdgcn_s_memtime();
#else
#ifndef VLG_ABL_NOBW
    dmv_bw_all<SR, LONGSPAN>(cb, tid, nt, x);
#endif
#endif
    }
    // expected counts out (coalesced; padded positions get exact zeros like the reference).  Decode mode
    // (heads != null): heads[c] = the h with a non-zero attach count -- what the callers compute on the host
    // with `argmax.sum(-1).nonzero()` + a scatter (ldndmv.py:301-303, joint.py:256-258); 0 for root / padding.
    io.clear_heads(tid, nt);
    x.sync();
    const int E = io.out_extent(Ne);
    const float inv_e = 1.0f / (float)E;
#ifdef VLG_ABL_NOOUT
    if (E > 0) return;
#endif
    for (int idx = tid; idx < E * E; idx += nt) {
        const int h = (int)(((float)idx + 0.5f) * inv_e), ch = idx - h * E;
        float2 g = oo;
        if (h < Ne && ch < Ne) {
            if (ch < h) g = c.gI[h * P + ch];
            else if (ch > h) g = c.gI[h * P + ch + 1];
        }
        io.st_attach(h, ch, g);
    }
    if (io.wants_dec()) {
        // STOP counts: the adjoint of the width-0 span (one cell each)
        for (int idx = tid; idx < E * 4; idx += nt) {
            const int h = idx >> 2, dir = (idx >> 1) & 1, v = idx & 1;
            float g = 0.f;
            if (h < Ne) {
                const int q = h * P + h + dir;
                g = SR == VLG_SR_MAX ? c.gdecs[h * 4 + dir * 2 + v] : (v == 1 ? c.gCc[q] : 0.f) + reinterpret_cast<const float*>(c.gCi + q)[v];
            }
            io.st_dec(h, (dir * 2 + v) * 2 + 1, g);
        }
        // GO counts: dec[h,dir,v,GO] enters every incomplete span headed by h towards dir (dmv.py:36-37), so its count is a sum
        // over a row of the finished gI chart.  Four lanes per (h, dir) take every fourth cell (both valences at once) and meet in
        // a two-step butterfly: a fixed summation tree, ~10 dependent LDS reads instead of the 40 of one lane per sum (this loop was
        // 2.2 us of the 77 us launch as a serial walk).
        constexpr int GL = 4;
        for (int base = 0; base < E * 2; base += nt / GL) {
            const int grp = base + tid / GL, rr = tid % GL;
            const int h = grp >> 1, dir = grp & 1;
            float sum[2] = {0.f, 0.f};
            if (grp < E * 2 && h < Ne) {
                const float2* row = c.gI + h * P + (dir == 0 ? 0 : h + 2);      // cells (h, ch) for ch < h  |  (h, ch + 1) for ch > h
                const int n = dir == 0 ? h : Ne - 1 - h;
                for (int k = rr; k < n; k += GL) { sum[0] += row[k].x; sum[1] += row[k].y; }
            }
            x.template allreduce_sum<2>(sum, GL);
            if (grp < E * 2 && rr == 0) {
                io.st_dec(h, (dir * 2 + 0) * 2, sum[0]);
                io.st_dec(h, (dir * 2 + 1) * 2, sum[1]);
            }
        }
    }
#if defined(VLG_STAMP) && defined(__HIPCC__)
    x.sync();
    if ((tid & 63) == 0) {   // diagnostic build only: per-wave cycle sums overwrite the (padded) last rows of grad_dec
        float* o = io.gdec + (size_t)(io.N - 1 - (tid >> 6)) * 8;
        o[0] = (float)st_body; o[1] = (float)st_sync; o[2] = (float)st_bbody; o[3] = (float)st_bsync;
        o[4] = (float)(st_end - st_t0); o[5] = (float)st_stage; o[6] = 0.f; o[7] = 0.f;
        float* o2 = io.gdec + (size_t)(io.N - 1 - 8 - (tid >> 6)) * 8;
        for (int k = 0; k < 8; ++k) o2[k] = (float)x.acc[k];
    }
#endif
}

// DepTree: the same walk without valences (single back-pointer per cell).
VLG_HD void dep_walk(const DepCtx& c, float g) {
    const int P = c.P;
    int* stack = reinterpret_cast<int*>(c.gCc);
    int top = 0;
    int e = walk_entry(1, 0, c.len, 0);   // CR(0, len), deptree.py:74-75
    for (;;) {
        const int kind = e & 3, a = (e >> 2) & 255, b = (e >> 10) & 255;
        if (kind < 2 && a == b) {
            if (top == 0) break;
            e = stack[--top];
            continue;
        }
        int sib;
        if (kind == 0) {
            const int r = c.bpC[a * P + b];
            sib = walk_entry(0, b + r, b, 0);
            e = walk_entry(2, a, b + r, 0);
        } else if (kind == 1) {
            const int r = c.bpC[a * P + b + 1];
            sib = walk_entry(1, a + 1 + r, b, 0);
            e = walk_entry(3, a, a + 1 + r, 0);
        } else if (kind == 2) {
            c.gI[a * P + b] = g;
            const int r = c.bpS[b * P + a];   // IL(j,k) and IR(k,j) share T(k,j), kept once at [left end][right end]
            sib = walk_entry(1, b, b + r, 0);
            e = walk_entry(0, a, b + r + 1, 0);
        } else {
            c.gI[a * P + b + 1] = g;
            const int r = c.bpS[a * P + b];
            sib = walk_entry(1, a, a + r, 0);
            e = walk_entry(0, b, a + r + 1, 0);
        }
        stack[top++] = sib;
    }
}

template <int SR, bool BWD, typename In, int SPANS = kSpansGeneral, typename X>
VLG_HD void dep_run(const DepCtx& c, const typename In::T* arc, int N, float glogZ, float* logZ, float* garc,
                    long long* heads, int tid, int nt, X& x) {
    const int Ne = c.Ne, P = c.P, len = c.len;
    for (int i = tid; i < Ne * P; i += nt) {
        c.C[i] = VLG_NEGINF;   // deptree.py:42-43
        c.I[i] = VLG_NEGINF;
        if (BWD) {
            c.gI[i] = 0.f;
            if (SR != VLG_SR_MAX) { c.gCc[i] = 0.f; c.gCi[i] = 0.f; }   // Max: gCc is only the walk's stack
        }
    }
    x.sync();
    // arcs beyond the sentence are never read (the reference masks them on a clone, deptree.py:159-161)
    for (int idx = tid; idx < Ne * Ne; idx += nt) {
        const int h = idx / Ne, ch = idx - h * Ne;
        if (ch == h) {
            c.C[h * P + h] = 0.f;        // semiring one, deptree.py:44
            c.C[h * P + h + 1] = 0.f;
        } else {
            const float a = pot_clamp(In::ld(arc, (size_t)h * N + ch)) * VLG_LOG2E;
            if (ch < h) c.I[h * P + ch] = a;
            else c.I[h * P + ch + 1] = a;
        }
    }
    x.sync();
    dep_fw_all<SR, BWD, SPANS>(c, tid, nt, x);
    if (tid == 0) *logZ = c.C[len + 1] * VLG_LN2;   // CR(0,len), deptree.py:74-75
    if (!BWD) return;
    if (SR == VLG_SR_MAX) {   // the best tree's arcs are all the Max semiring's gradient is: walk the back-pointers
        if (tid == 0) dep_walk(c, glogZ);
        x.sync();
    } else {
        if (tid == 0) c.gCc[len + 1] = glogZ;
        x.sync();
        dep_bw_all<SR, SPANS>(c, tid, nt, x);
    }
    if (heads) {
        for (int i = tid; i < N; i += nt) heads[i] = 0;
        x.sync();
    }
    for (int idx = tid; idx < N * N; idx += nt) {
        const int h = idx / N, ch = idx - h * N;
        float g = 0.f;
        if (h < Ne && ch < Ne && ch != h) {
            g = ch < h ? c.gI[h * P + ch] : c.gI[h * P + ch + 1];
            if (SR != VLG_SR_MAX) {   // + the self term the sweep left out (dep_self)
                const int i = ch < h ? ch : h, w = ch < h ? h - ch : ch - h;
                g += dep_self<SR>(c, i * (P + 1), w, ch < h ? 0 : 1);
            }
        }
        if (garc) garc[idx] = g;
        if (heads && g != 0.f) heads[ch] = h;
    }
}

// ---- byte layout of one sentence's working set, shared by the LDS carve, the global workspace and
//      the host phase emulator (which allocates EXACTLY this, with canaries behind it).
// Placement modes (what lives in LDS; the rest goes to the caller's workspace, L2-resident):
//   0: everything          1: adjoints + staging (value charts in workspace)
//   2: gCc, gCi + staging  3: staging only
VLG_HOSTDEV size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct Region {
    size_t off;
    bool lds;
};

struct Carver {
    size_t lds = 0, ws = 0;
    VLG_HOSTDEV_M Region take(size_t bytes, bool in_lds) {
        Region r;
        r.lds = in_lds;
        if (in_lds) { r.off = lds; lds = align16(lds + bytes); }
        else { r.off = ws; ws = align16(ws + bytes); }
        return r;
    }
};

struct DmvLayout {
    Region C, I, S, bpS, bpC, gCc, gCi, gI, decs, gdecs;
    Region C_in, I_in;   // the value charts as the INSIDE pass sees them (== C, I except in the overlay mode below)
    size_t lds_bytes, ws_bytes;
    // walk: Max semiring with an outside pass (tree, counts or both) -- no replay tape (S), no adjoint charts; gCc shrinks to the
    // walk's stack, gI stays (the one-hot attach output / head vector / GO counts are read from it).  Its modes keep the value charts and
    // back-pointers in LDS longest:  0: everything   1: gI in the workspace   2+: value charts too.
    VLG_HOSTDEV_M DmvLayout(int N, bool bwd, bool is_max, int mode, bool walk = false) {
        const size_t cells = (size_t)N * chart_pitch(N);
        Carver k;
        if (walk) {
            const bool v = mode < 2, a2 = mode < 1;
            C = k.take(cells * 8, v);
            I = k.take(cells * 8, v);
            S = k.take(0, v);
            bpS = k.take(cells, v);
            bpC = k.take(cells * 2, v);
            gCc = k.take((2 * (size_t)N + 4) * 4, true);   // the walk's stack: one pending sibling per open span, depth <= 2N
            gCi = k.take(0, true);
            gI = k.take(cells * 8, a2);
            decs = k.take((size_t)N * 32, true);
            gdecs = k.take((size_t)N * 16, true);            // the walk's STOP counts [N][dir][val]
            C_in = C;
            I_in = I;
        } else {
            const bool v = mode < 1, a1 = mode < 3, a2 = mode < 2;
            // mode 1 with an outside pass: the LDS cannot hold values AND adjoints, but it can hold either.  The inside
            // pass keeps C, I in LDS (its inner loop never touches global memory; only the replay tape S and the
            // back-pointers are streamed out), then the two charts are copied to the workspace in one coalesced sweep
            // and the adjoint charts are laid over them for the outside pass, which reads values from the workspace.
            const bool overlay = bwd && mode == 1;
            if (overlay) {   // adjoints first: they start at LDS offset 0, so do the overlaid inside-pass charts
                gCc = k.take(cells * 4, true);
                gCi = k.take(cells * 8, true);
                gI = k.take(cells * 8, true);
                C_in.lds = true; C_in.off = 0;
                I_in.lds = true; I_in.off = align16(cells * 8);   // 16 cells <= the 20 cells of adjoints
            }
            C = k.take(cells * 8, v);
            I = k.take(cells * 8, v);
            S = k.take(bwd ? cells * 4 : 0, v);
            bpS = k.take(bwd && is_max ? cells : 0, v);
            bpC = k.take(bwd && is_max ? cells * 2 : 0, v);
            if (!overlay) {
                gCc = k.take(bwd ? cells * 4 : 0, a1);
                gCi = k.take(bwd ? cells * 8 : 0, a1);
                gI = k.take(bwd ? cells * 8 : 0, a2);
                C_in = C;
                I_in = I;
            }
            decs = k.take((size_t)N * 32, true);
            gdecs = k.take(0, true);   // (dec[...,GO] counts are summed from the finished gI chart in the output stage)
        }
        lds_bytes = k.lds;
        ws_bytes = k.ws;
    }
};

struct DepLayout {
    Region C, I, S, bpS, bpC, gCc, gCi, gI;
    size_t lds_bytes, ws_bytes;
    // Max semiring + gradient = always the back-pointer walk (see DmvLayout's walk variant)
    VLG_HOSTDEV_M DepLayout(int N, bool bwd, bool is_max, int mode, bool = false) {
        const size_t cells = (size_t)N * chart_pitch(N);
        Carver k;
        if (bwd && is_max) {
            const bool v = mode < 2, a2 = mode < 1;
            C = k.take(cells * 4, v);
            I = k.take(cells * 4, v);
            S = k.take(0, v);
            bpS = k.take(cells, v);
            bpC = k.take(cells, v);
            gCc = k.take((2 * (size_t)N + 4) * 4, true);   // the walk's stack: one pending sibling per open span, depth <= 2N
            gCi = k.take(0, true);
            gI = k.take(cells * 4, a2);
        } else {
            const bool v = mode < 1, a1 = mode < 3, a2 = mode < 2;
            C = k.take(cells * 4, v);
            I = k.take(cells * 4, v);
            S = k.take(bwd ? cells * 4 : 0, v);
            bpS = k.take(0, v);
            bpC = k.take(0, v);
            gCc = k.take(bwd ? cells * 4 : 0, a1);
            gCi = k.take(bwd ? cells * 4 : 0, a1);
            gI = k.take(bwd ? cells * 4 : 0, a2);
        }
        lds_bytes = k.lds;
        ws_bytes = k.ws;
    }
};

template <typename Layout>
VLG_HOSTDEV int pick_mode(int N, bool bwd, bool is_max, size_t lds_budget, bool walk = false) {
    for (int mode = 0; mode < 3; ++mode)
        if (Layout(N, bwd, is_max, mode, walk).lds_bytes <= lds_budget) return mode;
    return 3;
}

}  // namespace vlg
