// vlg_dp_core.h -- per-thread phase bodies of the structured-DP kernels (DMV1o and DepTree).
//
// Each function is the work of ONE thread (tid of nt) in ONE barrier-delimited phase of the
// span-width loop.  The HIP kernels (vlg_dp.hip) call them between __syncthreads(); the
// host-side phase emulator used by the CPU test-suite (tests/emu/emu_dp.cpp) calls the very
// same bodies for tid = 0..nt-1 (in both orders, to expose intra-phase races).  Nothing here
// is a CPU fallback: the product only ever runs these through the gfx950 kernels.
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
// Outside: every cell is written once, so the charts are the tape; the adjoint of
// out = lse_r t_r is t_r_bar += out_bar * exp(t_r - out) (Max semiring: the first arg-max only).
// Phase B1 distributes the adjoints of the complete spans of width w, phase B2 those of the
// incomplete spans; within a phase every read-modify-write target is owned by exactly one
// (span, r) pair, so no atomics are needed (ownership argument: DESIGN.md, "Outside pass").
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VLG_HD __device__ __forceinline__
#define VLG_HOSTDEV __host__ __device__ inline
#define VLG_HDM __device__ __forceinline__   // member-function form
#define VLG_HOSTDEV_M __host__ __device__
#define VLG_EXP(x) __expf(x)
#define VLG_LOG(x) __logf(x)
#else
#include <cmath>
#include <cstddef>
#include <cstdint>
#define VLG_HD static inline
#define VLG_HOSTDEV static inline
#define VLG_HDM inline
#define VLG_HOSTDEV_M
#define VLG_EXP(x) expf(x)
#define VLG_LOG(x) logf(x)
struct float2 { float x, y; };
static inline float2 make_float2(float a, float b) { float2 r; r.x = a; r.y = b; return r; }
#endif

#if defined(__HIPCC__)
#define VLG_BITS2F(u) __uint_as_float(u)
#else
#include <cstring>
static inline float vlg_bits2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
#define VLG_BITS2F(u) vlg_bits2f(u)
#endif

#define VLG_NEGINF (-1e12f)  // semiring zero, semirings.py:16,128 (finite sentinel, never -inf)
#define VLG_SR_LOG 0
#define VLG_SR_MAX 1

namespace vlg {

// ---- input element types: potentials arrive as fp32 or bf16; all arithmetic is fp32 ----------------
struct F32In {
    using T = float;
    static VLG_HDM float ld(const T* p, size_t i) { return p[i]; }
    static VLG_HDM float2 ld2(const T* p, size_t i) { return make_float2(p[i], p[i + 1]); }
};
struct BF16In {
    using T = uint16_t;
    static VLG_HDM float ld(const T* p, size_t i) { return VLG_BITS2F((uint32_t)p[i] << 16); }
    static VLG_HDM float2 ld2(const T* p, size_t i) {
        return make_float2(VLG_BITS2F((uint32_t)p[i] << 16), VLG_BITS2F((uint32_t)p[i + 1] << 16));
    }
};

// dec[h] is 8 floats [dir][val][decision]; helpers for the staged copy.
VLG_HD int dec_idx(int dir, int val, int z) { return (dir * 2 + val) * 2 + z; }

struct DmvCtx {
    int Ne;         // len + 1: only spans inside [0, len] are ever read by valid cells
    int len;
    int P;          // chart pitch (cells), odd, >= Ne + 1
    float2* C;      // complete spans   [Ne][P]
    float2* I;      // incomplete spans [Ne][P]  (pre-loaded with attach + dec[...,GO])
    float* S;       // SL(i,j) at S[j*P+i], SR(i,j) at S[i*P+j]
    float2* gC;     // adjoints
    float2* gI;
    float* decs;    // staged dec        [Ne][8]
    float* gdecs;   // adjoint of dec    [Ne][8]
    unsigned char* bpS;   // Max semiring back-pointers (first arg-max r), same indexing as S
    unsigned char* bpC;   // [Ne][P][2]
};

// ------------------------------------------------------------------------------------------------
// DMV1o phase F1(w): SL / SR and the incomplete spans.  2 * (Ne - w) work items.
// ------------------------------------------------------------------------------------------------
template <int SR, bool BWD>
VLG_HD void dmv_f1(const DmvCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w;
    for (int idx = tid; idx < 2 * n; idx += nt) {
        const int i = idx >> 1, side = idx & 1, j = i + w;
        const float2* cr = c.C + i * P + i + 1;   // CR(i, i+r)   at +r
        const float2* cl = c.C + j * P + i + 1;   // CL(j, i+r+1) at +r
        float m = -3.0e38f;
        int am = 0;
        for (int r = 0; r < w; ++r) {
            const float t = side == 0 ? cr[r].y + cl[r].x : cr[r].x + cl[r].y;
            if (t > m) { m = t; am = r; }
        }
        float out = m;
        if (SR == VLG_SR_LOG) {
            float s = 0.f;
            for (int r = 0; r < w; ++r) {
                const float t = side == 0 ? cr[r].y + cl[r].x : cr[r].x + cl[r].y;
                s += VLG_EXP(t - m);
            }
            out = m + VLG_LOG(s);
        }
        if (side == 0) {
            if (BWD) c.S[j * P + i] = out;
            if (BWD && SR == VLG_SR_MAX) c.bpS[j * P + i] = (unsigned char)am;
            float2 a = c.I[j * P + i];
            c.I[j * P + i] = make_float2(a.x + out, a.y + out);
        } else {
            if (BWD) c.S[i * P + j] = out;
            if (BWD && SR == VLG_SR_MAX) c.bpS[i * P + j] = (unsigned char)am;
            float2 a = c.I[i * P + j + 1];
            c.I[i * P + j + 1] = make_float2(a.x + out, a.y + out);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// DMV1o phase F2(w): complete spans.  4 * (Ne - w) work items (span, side, valence).
// ------------------------------------------------------------------------------------------------
template <int SR, bool BWD>
VLG_HD void dmv_f2(const DmvCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w;
    for (int idx = tid; idx < 4 * n; idx += nt) {
        const int i = idx >> 2, side = (idx >> 1) & 1, v = idx & 1, j = i + w;
        float m = -3.0e38f;
        int am = 0;
        float out;
        if (side == 0) {
            const float2* a = c.C + i * P + i;        // CL(i+r, i) at +r*P
            const float* b = (const float*)(c.I + j * P + i) + v;   // IL(j, i+r).v at +2r
            for (int r = 0; r < w; ++r) {
                const float t = a[r * P].y + b[2 * r];
                if (t > m) { m = t; am = r; }
            }
            out = m;
            if (SR == VLG_SR_LOG) {
                float s = 0.f;
                for (int r = 0; r < w; ++r) s += VLG_EXP(a[r * P].y + b[2 * r] - m);
                out = m + VLG_LOG(s);
            }
            ((float*)(c.C + j * P + i))[v] = out;
            if (BWD && SR == VLG_SR_MAX) c.bpC[(j * P + i) * 2 + v] = (unsigned char)am;
        } else {
            const float* a = (const float*)(c.I + i * P + i + 2) + v;   // IR(i, i+1+r).v at +2r
            const float2* b = c.C + (i + 1) * P + j + 1;                // CR(i+1+r, j) at +r*P
            for (int r = 0; r < w; ++r) {
                const float t = a[2 * r] + b[r * P].y;
                if (t > m) { m = t; am = r; }
            }
            out = m;
            if (SR == VLG_SR_LOG) {
                float s = 0.f;
                for (int r = 0; r < w; ++r) s += VLG_EXP(a[2 * r] + b[r * P].y - m);
                out = m + VLG_LOG(s);
            }
            if (i == 0 && w != c.len) out = VLG_NEGINF;   // single-root constraint, dmv.py:63
            ((float*)(c.C + i * P + j + 1))[v] = out;
            if (BWD && SR == VLG_SR_MAX) c.bpC[(i * P + j + 1) * 2 + v] = (unsigned char)am;
        }
    }
}

// weight of term r in a reduction with result `out`, scaled by the upstream adjoint g
template <int SR>
VLG_HD float adj_w(float g, float t, float out, int r, int bp) {
    if (SR == VLG_SR_MAX) return r == bp ? g : 0.f;
    return g != 0.f ? g * VLG_EXP(t - out) : 0.f;
}

// ------------------------------------------------------------------------------------------------
// DMV1o phase B1(w): adjoints of the complete spans of width w.  (Ne - w) * w work items.
// ------------------------------------------------------------------------------------------------
template <int SR>
VLG_HD void dmv_b1(const DmvCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w, total = n * w;
    const float inv_w = 1.0f / (float)w;
    for (int idx = tid; idx < total; idx += nt) {
        int i = (int)(((float)idx + 0.5f) * inv_w);
        int r = idx - i * w;
        const int j = i + w;
        // CL(j,i).v = (+)_r CL(i+r,i).NC + IL(j,i+r).v
        {
            const float2 g = c.gC[j * P + i];
            const float2 out = c.C[j * P + i];
            const float a = c.C[(i + r) * P + i].y;
            const float2 il = c.I[j * P + i + r];
            int bp0 = 0, bp1 = 0;
            if (SR == VLG_SR_MAX) { bp0 = c.bpC[(j * P + i) * 2]; bp1 = c.bpC[(j * P + i) * 2 + 1]; }
            const float w0 = adj_w<SR>(g.x, a + il.x, out.x, r, bp0);
            const float w1 = adj_w<SR>(g.y, a + il.y, out.y, r, bp1);
            float2 t = c.gI[j * P + i + r];
            c.gI[j * P + i + r] = make_float2(t.x + w0, t.y + w1);
            c.gC[(i + r) * P + i].y += w0 + w1;
        }
        // CR(i,j).v = (+)_r IR(i,i+1+r).v + CR(i+1+r,j).NC      (masked cell: no adjoint, dmv.py:63)
        {
            float2 g = c.gC[i * P + j + 1];
            if (i == 0 && w != c.len) g = make_float2(0.f, 0.f);
            const float2 out = c.C[i * P + j + 1];
            const float2 ir = c.I[i * P + i + r + 2];
            const float a = c.C[(i + 1 + r) * P + j + 1].y;
            int bp0 = 0, bp1 = 0;
            if (SR == VLG_SR_MAX) { bp0 = c.bpC[(i * P + j + 1) * 2]; bp1 = c.bpC[(i * P + j + 1) * 2 + 1]; }
            const float w0 = adj_w<SR>(g.x, ir.x + a, out.x, r, bp0);
            const float w1 = adj_w<SR>(g.y, ir.y + a, out.y, r, bp1);
            float2 t = c.gI[i * P + i + r + 2];
            c.gI[i * P + i + r + 2] = make_float2(t.x + w0, t.y + w1);
            c.gC[(i + 1 + r) * P + j + 1].y += w0 + w1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// DMV1o phase B2(w): adjoints of the incomplete spans of width w.  (Ne - w) * w work items.
// ------------------------------------------------------------------------------------------------
template <int SR>
VLG_HD void dmv_b2(const DmvCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w, total = n * w;
    const float inv_w = 1.0f / (float)w;
    for (int idx = tid; idx < total; idx += nt) {
        int i = (int)(((float)idx + 0.5f) * inv_w);
        int r = idx - i * w;
        const int j = i + w;
        const float2 gil = c.gI[j * P + i];        // total adjoint of IL(j,i).v == d logZ / d attach[j,i,v]
        const float2 gir = c.gI[i * P + j + 1];
        const float gsl = gil.x + gil.y, gsr = gir.x + gir.y;
        const float2 cr = c.C[i * P + i + r + 1];  // CR(i, i+r)
        const float2 cl = c.C[j * P + i + r + 1];  // CL(j, i+r+1)
        int bpl = 0, bpr = 0;
        if (SR == VLG_SR_MAX) { bpl = c.bpS[j * P + i]; bpr = c.bpS[i * P + j]; }
        const float wl = adj_w<SR>(gsl, cr.y + cl.x, c.S[j * P + i], r, bpl);   // SL term: CR.NC + CL.HC
        const float wr = adj_w<SR>(gsr, cr.x + cl.y, c.S[i * P + j], r, bpr);   // SR term: CR.HC + CL.NC
        float2 t = c.gC[i * P + i + r + 1];
        c.gC[i * P + i + r + 1] = make_float2(t.x + wr, t.y + wl);
        t = c.gC[j * P + i + r + 1];
        c.gC[j * P + i + r + 1] = make_float2(t.x + wl, t.y + wr);
        if (r == 0) {   // dec[h,dir,v,GO] enters every incomplete span headed by h (dmv.py:36-37)
            c.gdecs[j * 8 + dec_idx(0, 0, 0)] += gil.x;
            c.gdecs[j * 8 + dec_idx(0, 1, 0)] += gil.y;
            c.gdecs[i * 8 + dec_idx(1, 0, 0)] += gir.x;
            c.gdecs[i * 8 + dec_idx(1, 1, 0)] += gir.y;
        }
    }
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
    float* gC;
    float* gI;
    unsigned char* bpS;
    unsigned char* bpC;
};

template <int SR, bool BWD>
VLG_HD void dep_f1(const DepCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w;
    for (int i = tid; i < n; i += nt) {
        const int j = i + w;
        const float* cr = c.C + i * P + i + 1;
        const float* cl = c.C + j * P + i + 1;
        float m = -3.0e38f;
        int am = 0;
        for (int r = 0; r < w; ++r) {
            const float t = cr[r] + cl[r];
            if (t > m) { m = t; am = r; }
        }
        float out = m;
        if (SR == VLG_SR_LOG) {
            float s = 0.f;
            for (int r = 0; r < w; ++r) s += VLG_EXP(cr[r] + cl[r] - m);
            out = m + VLG_LOG(s);
        }
        if (BWD) c.S[i * P + j] = out;
        if (BWD && SR == VLG_SR_MAX) c.bpS[i * P + j] = (unsigned char)am;
        c.I[j * P + i] += out;
        c.I[i * P + j + 1] += out;
    }
}

template <int SR, bool BWD>
VLG_HD void dep_f2(const DepCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w;
    for (int idx = tid; idx < 2 * n; idx += nt) {
        const int i = idx >> 1, side = idx & 1, j = i + w;
        float m = -3.0e38f;
        int am = 0;
        float out;
        if (side == 0) {
            const float* a = c.C + i * P + i;
            const float* b = c.I + j * P + i;
            for (int r = 0; r < w; ++r) {
                const float t = a[r * P] + b[r];
                if (t > m) { m = t; am = r; }
            }
            out = m;
            if (SR == VLG_SR_LOG) {
                float s = 0.f;
                for (int r = 0; r < w; ++r) s += VLG_EXP(a[r * P] + b[r] - m);
                out = m + VLG_LOG(s);
            }
            c.C[j * P + i] = out;
            if (BWD && SR == VLG_SR_MAX) c.bpC[j * P + i] = (unsigned char)am;
        } else {
            const float* a = c.I + i * P + i + 2;
            const float* b = c.C + (i + 1) * P + j + 1;
            for (int r = 0; r < w; ++r) {
                const float t = a[r] + b[r * P];
                if (t > m) { m = t; am = r; }
            }
            out = m;
            if (SR == VLG_SR_LOG) {
                float s = 0.f;
                for (int r = 0; r < w; ++r) s += VLG_EXP(a[r] + b[r * P] - m);
                out = m + VLG_LOG(s);
            }
            if (i == 0 && w != c.len) out = VLG_NEGINF;   // deptree.py:71-72
            c.C[i * P + j + 1] = out;
            if (BWD && SR == VLG_SR_MAX) c.bpC[i * P + j + 1] = (unsigned char)am;
        }
    }
}

template <int SR>
VLG_HD void dep_b1(const DepCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w, total = n * w;
    const float inv_w = 1.0f / (float)w;
    for (int idx = tid; idx < total; idx += nt) {
        int i = (int)(((float)idx + 0.5f) * inv_w);
        int r = idx - i * w;
        const int j = i + w;
        {
            const float g = c.gC[j * P + i];
            const float t = c.C[(i + r) * P + i] + c.I[j * P + i + r];
            const float wt = adj_w<SR>(g, t, c.C[j * P + i], r, SR == VLG_SR_MAX ? c.bpC[j * P + i] : 0);
            c.gI[j * P + i + r] += wt;
            c.gC[(i + r) * P + i] += wt;
        }
        {
            float g = c.gC[i * P + j + 1];
            if (i == 0 && w != c.len) g = 0.f;
            const float t = c.I[i * P + i + r + 2] + c.C[(i + 1 + r) * P + j + 1];
            const float wt = adj_w<SR>(g, t, c.C[i * P + j + 1], r, SR == VLG_SR_MAX ? c.bpC[i * P + j + 1] : 0);
            c.gI[i * P + i + r + 2] += wt;
            c.gC[(i + 1 + r) * P + j + 1] += wt;
        }
    }
}

template <int SR>
VLG_HD void dep_b2(const DepCtx& c, int w, int tid, int nt) {
    const int P = c.P, n = c.Ne - w, total = n * w;
    const float inv_w = 1.0f / (float)w;
    for (int idx = tid; idx < total; idx += nt) {
        int i = (int)(((float)idx + 0.5f) * inv_w);
        int r = idx - i * w;
        const int j = i + w;
        const float gs = c.gI[j * P + i] + c.gI[i * P + j + 1];
        const float t = c.C[i * P + i + r + 1] + c.C[j * P + i + r + 1];
        const float wt = adj_w<SR>(gs, t, c.S[i * P + j], r, SR == VLG_SR_MAX ? c.bpS[i * P + j] : 0);
        c.gC[i * P + i + r + 1] += wt;
        c.gC[j * P + i + r + 1] += wt;
    }
}

// ================================================================================================
// Whole-sentence drivers: the body of one workgroup.  `sync()` is __syncthreads() on the GPU and
// the token barrier of the host phase emulator in the CPU tests.  Pointers in the context are
// already carved (LDS and/or workspace); dec/attach/gdec/gatt/logZ point at THIS sentence.
// ================================================================================================
template <int SR, bool BWD, typename In, typename Sync>
VLG_HD void dmv_run(const DmvCtx& c, const typename In::T* dec, const typename In::T* attach, int N, float glogZ,
                    float* logZ, float* gdec, float* gatt, int tid, int nt, Sync sync) {
    const int Ne = c.Ne, P = c.P, len = c.len;
    // ---- stage: charts to the semiring zero (dmv.py:34-35), dec into fast memory -----------------
    const float2 zz = make_float2(VLG_NEGINF, VLG_NEGINF), oo = make_float2(0.f, 0.f);
    for (int i = tid; i < Ne * P; i += nt) {
        c.C[i] = zz;
        c.I[i] = zz;
        if (BWD) { c.gC[i] = oo; c.gI[i] = oo; }
    }
    for (int i = tid; i < Ne * 8; i += nt) {
        c.decs[i] = In::ld(dec, i);
        if (BWD) c.gdecs[i] = 0.f;
    }
    sync();
    // incomplete-span slots are pre-loaded with attach + dec[...,GO] (dmv.py:36-37); the width-0
    // complete spans with the STOP scores (dmv.py:39-40).  This folds the reference's
    // attach_left / attach_right temporaries into the load stage.
    for (int idx = tid; idx < Ne * Ne; idx += nt) {
        const int h = idx / Ne, ch = idx - h * Ne;
        const float* d = c.decs + h * 8;
        if (ch == h) {
            c.C[h * P + h] = make_float2(d[1], d[3]);        // CL(h,h).v = dec[h,LEFT ,v,STOP]
            c.C[h * P + h + 1] = make_float2(d[5], d[7]);    // CR(h,h).v = dec[h,RIGHT,v,STOP]
        } else {
            const float2 a = In::ld2(attach, ((size_t)h * N + ch) * 2);
            if (ch < h) c.I[h * P + ch] = make_float2(a.x + d[0], a.y + d[2]);
            else c.I[h * P + ch + 1] = make_float2(a.x + d[4], a.y + d[6]);
        }
    }
    sync();
    // ---- inside -----------------------------------------------------------------------------------
    for (int w = 1; w < Ne; ++w) {
        dmv_f1<SR, BWD>(c, w, tid, nt);
        sync();
        dmv_f2<SR, BWD>(c, w, tid, nt);
        sync();
    }
    if (tid == 0) *logZ = c.C[len + 1].y;   // CR(0,len).NOCHILD, dmv.py:65
    if (!BWD) return;
    // ---- outside: adjoint replay ------------------------------------------------------------------
    if (tid == 0) c.gC[len + 1].y = glogZ;
    sync();
    for (int w = Ne - 1; w >= 1; --w) {
        dmv_b1<SR>(c, w, tid, nt);
        sync();
        dmv_b2<SR>(c, w, tid, nt);
        sync();
    }
    // expected counts out (coalesced; padded positions get exact zeros like the reference)
    for (int idx = tid; idx < N * N; idx += nt) {
        const int h = idx / N, ch = idx - h * N;
        float2 g = oo;
        if (h < Ne && ch < Ne) {
            if (ch < h) g = c.gI[h * P + ch];
            else if (ch > h) g = c.gI[h * P + ch + 1];
        }
        *reinterpret_cast<float2*>(gatt + (size_t)idx * 2) = g;
    }
    for (int idx = tid; idx < N * 8; idx += nt) {
        const int h = idx >> 3, k = idx & 7;
        float g = 0.f;
        if (h < Ne) {
            const int dir = k >> 2, v = (k >> 1) & 1;
            if ((k & 1) == 0) g = c.gdecs[h * 8 + k];                                   // GO
            else g = reinterpret_cast<const float*>(c.gC + h * P + h + dir)[v];         // STOP = width-0 span
        }
        gdec[idx] = g;
    }
}

template <int SR, bool BWD, typename In, typename Sync>
VLG_HD void dep_run(const DepCtx& c, const typename In::T* arc, int N, float glogZ, float* logZ, float* garc, int tid,
                    int nt, Sync sync) {
    const int Ne = c.Ne, P = c.P, len = c.len;
    for (int i = tid; i < Ne * P; i += nt) {
        c.C[i] = VLG_NEGINF;   // deptree.py:42-43
        c.I[i] = VLG_NEGINF;
        if (BWD) { c.gC[i] = 0.f; c.gI[i] = 0.f; }
    }
    sync();
    // arcs beyond the sentence are never read (the reference masks them on a clone, deptree.py:159-161)
    for (int idx = tid; idx < Ne * Ne; idx += nt) {
        const int h = idx / Ne, ch = idx - h * Ne;
        if (ch == h) {
            c.C[h * P + h] = 0.f;        // semiring one, deptree.py:44
            c.C[h * P + h + 1] = 0.f;
        } else {
            const float a = In::ld(arc, (size_t)h * N + ch);
            if (ch < h) c.I[h * P + ch] = a;
            else c.I[h * P + ch + 1] = a;
        }
    }
    sync();
    for (int w = 1; w < Ne; ++w) {
        dep_f1<SR, BWD>(c, w, tid, nt);
        sync();
        dep_f2<SR, BWD>(c, w, tid, nt);
        sync();
    }
    if (tid == 0) *logZ = c.C[len + 1];   // CR(0,len), deptree.py:74-75
    if (!BWD) return;
    if (tid == 0) c.gC[len + 1] = glogZ;
    sync();
    for (int w = Ne - 1; w >= 1; --w) {
        dep_b1<SR>(c, w, tid, nt);
        sync();
        dep_b2<SR>(c, w, tid, nt);
        sync();
    }
    for (int idx = tid; idx < N * N; idx += nt) {
        const int h = idx / N, ch = idx - h * N;
        float g = 0.f;
        if (h < Ne && ch < Ne) {
            if (ch < h) g = c.gI[h * P + ch];
            else if (ch > h) g = c.gI[h * P + ch + 1];
        }
        garc[idx] = g;
    }
}

// chart pitch: odd and >= N + 1 so that row-strided (column) walks hit distinct LDS banks
VLG_HOSTDEV int chart_pitch(int N) { return (N + 1) | 1; }

// ---- byte layout of one sentence's working set, shared by the LDS carve, the global workspace and
//      the host phase emulator (which allocates EXACTLY this, with canaries behind it) ------------------
VLG_HOSTDEV size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct DmvLayout {
    size_t C, I, S, bpS, bpC, gC, gI, decs, gdecs, value_end, total;
    VLG_HOSTDEV_M DmvLayout(int N, bool bwd, bool is_max) {
        const size_t cells = (size_t)N * chart_pitch(N);
        size_t o = 0;
        C = o; o = align16(o + cells * 8);
        I = o; o = align16(o + cells * 8);
        S = o; o = align16(o + (bwd ? cells * 4 : 0));
        bpS = o; o = align16(o + (bwd && is_max ? cells : 0));
        bpC = o; o = align16(o + (bwd && is_max ? cells * 2 : 0));
        value_end = o;
        gC = o; o = align16(o + (bwd ? cells * 8 : 0));
        gI = o; o = align16(o + (bwd ? cells * 8 : 0));
        decs = o; o = align16(o + (size_t)N * 32);
        gdecs = o; o = align16(o + (bwd ? (size_t)N * 32 : 0));
        total = o;
    }
};

struct DepLayout {
    size_t C, I, S, bpS, bpC, gC, gI, value_end, total;
    VLG_HOSTDEV_M DepLayout(int N, bool bwd, bool is_max) {
        const size_t cells = (size_t)N * chart_pitch(N);
        size_t o = 0;
        C = o; o = align16(o + cells * 4);
        I = o; o = align16(o + cells * 4);
        S = o; o = align16(o + (bwd ? cells * 4 : 0));
        bpS = o; o = align16(o + (bwd && is_max ? cells : 0));
        bpC = o; o = align16(o + (bwd && is_max ? cells : 0));
        value_end = o;
        gC = o; o = align16(o + (bwd ? cells * 4 : 0));
        gI = o; o = align16(o + (bwd ? cells * 4 : 0));
        total = o;
    }
};

}  // namespace vlg
