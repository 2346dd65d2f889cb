// vlg_dp_kernels.h -- device side of the batched structured DP: lane-exchange primitives, the __global__ kernel
// templates (DMV1o from merged potentials / from rule tables, DepTree) and their launchers.  Included by vlg_dp.hip
// (C ABI, argument checks, the small element-wise kernels) and by vlg_dp_inst.hip, which the build compiles once per
// (family, semiring, input type): the ~100 kernel instantiations are 12 translation units that build in parallel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_dp_core.h"

namespace vlg {

#ifndef VLG_DP_THREADS
#define VLG_DP_THREADS 512   // eight wavefronts, two per SIMD.  (Round 4 measured 640 / 768 threads with a 320 / 384-lane outside pass: the single
                             //  launch gains 1.5 us, but ten wavefronts of ~120 VGPRs no longer let a second DP workgroup share the CU -- the
                             //  marginals || Viterbi pair of lang_feat_max_tree went from 101 to 145 us -- so it stays 512.)
#endif
constexpr int kThreads = VLG_DP_THREADS;    // lanes per sentence (workgroup size)
constexpr size_t kLdsBudget = 160 * 1024;   // CDNA4 LDS per CU / per workgroup

// ---- cross-lane exchange: lane l <- lane l ^ K, for values that are uniform over aligned K-blocks ------
// (true at every step of an ascending butterfly all-reduce).  K = 1, 2 are quad permutes; K = 4 / 8 use
// the DPP half-row / row mirrors (the partner block's value is uniform, so any lane of it will do);
// K = 16 is a bit-mode ds_swizzle inside each 32-lane half; K = 32 goes through ds_bpermute.
template <int K>
__device__ __forceinline__ int xlane_i(int v) {
    if (K == 1) return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
    if (K == 2) return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
    if (K == 4) return __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false);   // row_half_mirror
    if (K == 8) return __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false);   // row_mirror
    if (K == 16) return __builtin_amdgcn_ds_swizzle(v, 0x401F);                     // and 0x1F, or 0, xor 0x10
    return __shfl_xor(v, 32, 64);
}
template <int K>
__device__ __forceinline__ float xlane(float v) { return __int_as_float(xlane_i<K>(__float_as_int(v))); }

// Fused DPP reduction steps: dst = op(dst, lane-permuted dst) in ONE instruction per value.  hipcc does not
// fold v_mov_b32_dpp into the consuming VALU op here, so a whole butterfly step (all n values) is one inline
// asm block; its leading s_nop covers the "VALU write -> DPP read" wait states that the assembler does not
// insert for asm.  Steps 16 and 32 use gfx950's v_permlane16_swap / v_permlane32_swap: with both operands
// holding x, the two results are (own, partner) in some order on every lane, so op(r0, r1) is the step --
// no LDS round trip (ds_swizzle / ds_bpermute cost a full LDS latency per value).
#define VLG_DPP_QP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
#define VLG_DPP_QP2 "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
#define VLG_DPP_HM "row_half_mirror row_mask:0xf bank_mask:0xf"
#define VLG_DPP_RM "row_mirror row_mask:0xf bank_mask:0xf"
#define VLG_DPP1(OP, C) "\n\t" OP " %0, %0, %0 " C
#define VLG_DPP2(OP, C) VLG_DPP1(OP, C) "\n\t" OP " %1, %1, %1 " C
#define VLG_DPP3(OP, C) VLG_DPP2(OP, C) "\n\t" OP " %2, %2, %2 " C
#define VLG_DPP4(OP, C) VLG_DPP3(OP, C) "\n\t" OP " %3, %3, %3 " C
#define VLG_DPP6(OP, C) VLG_DPP4(OP, C) "\n\t" OP " %4, %4, %4 " C "\n\t" OP " %5, %5, %5 " C

// one butterfly step over n registers of type T with DPP control string C
#define VLG_DPP_STEP(OP, C, v, n)                                                                                    \
    do {                                                                                                             \
        if constexpr ((n) == 6) asm("s_nop 1" VLG_DPP6(OP, C) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); \
        else if constexpr ((n) == 4) asm("s_nop 1" VLG_DPP4(OP, C) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); \
        else if constexpr ((n) == 3) asm("s_nop 1" VLG_DPP3(OP, C) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));          \
        else if constexpr ((n) == 2) asm("s_nop 1" VLG_DPP2(OP, C) : "+v"(v[0]), "+v"(v[1]));                       \
        else {                                                                                                       \
            _Pragma("unroll") for (int k = 0; k < (n); ++k) asm("s_nop 1" VLG_DPP1(OP, C) : "+v"(v[k]));           \
        }                                                                                                            \
    } while (0)

enum { kOpMax = 0, kOpAdd = 1, kOpMinI = 2 };

template <int OPK, int K, int n, typename T>
__device__ __forceinline__ void butterfly_step(T* v) {
    if constexpr (K <= 8) {
        if constexpr (OPK == kOpMax) {
            if constexpr (K == 1) VLG_DPP_STEP("v_max_f32_dpp", VLG_DPP_QP1, v, n);
            else if constexpr (K == 2) VLG_DPP_STEP("v_max_f32_dpp", VLG_DPP_QP2, v, n);
            else if constexpr (K == 4) VLG_DPP_STEP("v_max_f32_dpp", VLG_DPP_HM, v, n);
            else VLG_DPP_STEP("v_max_f32_dpp", VLG_DPP_RM, v, n);
        } else if constexpr (OPK == kOpAdd) {
            if constexpr (K == 1) VLG_DPP_STEP("v_add_f32_dpp", VLG_DPP_QP1, v, n);
            else if constexpr (K == 2) VLG_DPP_STEP("v_add_f32_dpp", VLG_DPP_QP2, v, n);
            else if constexpr (K == 4) VLG_DPP_STEP("v_add_f32_dpp", VLG_DPP_HM, v, n);
            else VLG_DPP_STEP("v_add_f32_dpp", VLG_DPP_RM, v, n);
        } else {
            if constexpr (K == 1) VLG_DPP_STEP("v_min_i32_dpp", VLG_DPP_QP1, v, n);
            else if constexpr (K == 2) VLG_DPP_STEP("v_min_i32_dpp", VLG_DPP_QP2, v, n);
            else if constexpr (K == 4) VLG_DPP_STEP("v_min_i32_dpp", VLG_DPP_HM, v, n);
            else VLG_DPP_STEP("v_min_i32_dpp", VLG_DPP_RM, v, n);
        }
    } else {
#pragma unroll
        for (int k = 0; k < n; ++k) {
            unsigned u;
            if constexpr (sizeof(T) == 4 && OPK == kOpMinI) u = (unsigned)v[k];
            else u = __float_as_uint((float)v[k]);
            const auto r = K == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false)
                                   : __builtin_amdgcn_permlane32_swap(u, u, false, false);
            if constexpr (OPK == kOpMax) v[k] = (T)fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
            else if constexpr (OPK == kOpAdd) v[k] = (T)(__uint_as_float(r[0]) + __uint_as_float(r[1]));
            else v[k] = (T)min((int)r[0], (int)r[1]);
        }
    }
}

// The DPP steps of one butterfly (lane distances 1, 2, 4, 8) as ONE asm block, for n >= 3 values: between a value's step and its next
// step lie the n - 1 >= 2 DPP instructions of the other values, which is the two wait states "VALU write -> DPP read" asks for -- so
// only the block's leading s_nop is needed (one block per step paid s_nop 1 per step, plus the s_nop 0 hipcc puts behind every asm).
#define VLG_DPP_FUSED(OP, NSTEPS, v, n)                                                                                              \
    do {                                                                                                                             \
        if constexpr ((n) == 3) {                                                                                                    \
            if constexpr ((NSTEPS) == 2) asm("s_nop 1" VLG_DPP3(OP, VLG_DPP_QP1) VLG_DPP3(OP, VLG_DPP_QP2) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])); \
            else if constexpr ((NSTEPS) == 3) asm("s_nop 1" VLG_DPP3(OP, VLG_DPP_QP1) VLG_DPP3(OP, VLG_DPP_QP2) VLG_DPP3(OP, VLG_DPP_HM) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])); \
            else asm("s_nop 1" VLG_DPP3(OP, VLG_DPP_QP1) VLG_DPP3(OP, VLG_DPP_QP2) VLG_DPP3(OP, VLG_DPP_HM) VLG_DPP3(OP, VLG_DPP_RM) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])); \
        } else if constexpr ((n) == 4) {                                                                                             \
            if constexpr ((NSTEPS) == 2) asm("s_nop 1" VLG_DPP4(OP, VLG_DPP_QP1) VLG_DPP4(OP, VLG_DPP_QP2) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); \
            else if constexpr ((NSTEPS) == 3) asm("s_nop 1" VLG_DPP4(OP, VLG_DPP_QP1) VLG_DPP4(OP, VLG_DPP_QP2) VLG_DPP4(OP, VLG_DPP_HM) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); \
            else asm("s_nop 1" VLG_DPP4(OP, VLG_DPP_QP1) VLG_DPP4(OP, VLG_DPP_QP2) VLG_DPP4(OP, VLG_DPP_HM) VLG_DPP4(OP, VLG_DPP_RM) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])); \
        } else {                                                                                                                     \
            if constexpr ((NSTEPS) == 2) asm("s_nop 1" VLG_DPP6(OP, VLG_DPP_QP1) VLG_DPP6(OP, VLG_DPP_QP2) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); \
            else if constexpr ((NSTEPS) == 3) asm("s_nop 1" VLG_DPP6(OP, VLG_DPP_QP1) VLG_DPP6(OP, VLG_DPP_QP2) VLG_DPP6(OP, VLG_DPP_HM) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); \
            else asm("s_nop 1" VLG_DPP6(OP, VLG_DPP_QP1) VLG_DPP6(OP, VLG_DPP_QP2) VLG_DPP6(OP, VLG_DPP_HM) VLG_DPP6(OP, VLG_DPP_RM) : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5])); \
        }                                                                                                                            \
    } while (0)

template <int OPK, int NSTEPS, int n, typename T>
__device__ __forceinline__ void butterfly_dpp_fused(T* v) {
    if constexpr (OPK == kOpMax) VLG_DPP_FUSED("v_max_f32_dpp", NSTEPS, v, n);
    else if constexpr (OPK == kOpAdd) VLG_DPP_FUSED("v_add_f32_dpp", NSTEPS, v, n);
    else VLG_DPP_FUSED("v_min_i32_dpp", NSTEPS, v, n);
}

template <int OPK, int n, typename T>
__device__ __forceinline__ void butterfly(T* v, int G) {   // G is uniform over the workgroup: no divergence
#ifndef VLG_NO_FUSED_BUTTERFLY
    if constexpr (n == 3 || n == 4 || n == 6) {
        if (G >= 4) {   // (G is a compile-time constant wherever a segment function inlines this)
            if (G == 4) butterfly_dpp_fused<OPK, 2, n>(v);
            else if (G == 8) butterfly_dpp_fused<OPK, 3, n>(v);
            else butterfly_dpp_fused<OPK, 4, n>(v);
            if (G > 16) butterfly_step<OPK, 16, n>(v);
            if (G > 32) butterfly_step<OPK, 32, n>(v);
            return;
        }
    }
#endif
    if (G > 1) butterfly_step<OPK, 1, n>(v);
    if (G > 2) butterfly_step<OPK, 2, n>(v);
    if (G > 4) butterfly_step<OPK, 4, n>(v);
    if (G > 8) butterfly_step<OPK, 8, n>(v);
    if (G > 16) butterfly_step<OPK, 16, n>(v);
    if (G > 32) butterfly_step<OPK, 32, n>(v);
}

struct DevX {
    static constexpr bool kSkipDeadWaves = true;   // the all-reduces are wave-local: a wave without spans can skip a phase
    static constexpr bool kChartsInLds = true;     // (with LONGSPAN = false) chart reads cannot fault: see dmv_fw_span's NOCLAMP
    __device__ __forceinline__ void sync() { __syncthreads(); }
    // The lanes of one lane group never straddle a wavefront and a wavefront executes its instructions in order for all
    // lanes at once, so within a group "all loads above, all stores below" needs no instruction.  (The host phase
    // emulator, whose lanes are free-running threads, makes this point a barrier.)
    __device__ __forceinline__ void lockstep() {}
    // a value the caller knows to be wave-uniform: pin it to an SGPR so that branches on it are scalar branches
    __device__ __forceinline__ bool uniform(bool v) { return __builtin_amdgcn_readfirstlane((int)v) != 0; }
    __device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#ifdef VLG_STAMP
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last = 0;
    __device__ __forceinline__ void stamp(int k) {   // acc[k & 7] += cycles since the previous stamp
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        acc[k & 7] += t - last;
        last = t;
    }
#endif
    template <int n>
    __device__ __forceinline__ void allreduce_max(float* v, int G) { butterfly<kOpMax, n>(v, G); }
    template <int n>
    __device__ __forceinline__ void allreduce_sum(float* v, int G) { butterfly<kOpAdd, n>(v, G); }
    // v[0], v[1] are held by lane `src01` of every group of G lanes, v[2], v[3] by lane `src23` (zero elsewhere): hand them
    // to the whole group.  Four ds_bpermute instead of 4 log2 G butterfly adds.
    __device__ __forceinline__ void group_bcast2(float* v, int G, int src0, int src1) {
        const int base = (int)(threadIdx.x & 63) & ~(G - 1);
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((base | src0) << 2, __builtin_bit_cast(int, v[0])));
        v[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((base | src1) << 2, __builtin_bit_cast(int, v[1])));
    }
    __device__ __forceinline__ void group_bcast4(float* v, int G, int src01, int src23) {
        const int base = (int)(threadIdx.x & 63) & ~(G - 1);
        const int a01 = (base | src01) << 2, a23 = (base | src23) << 2;
        v[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a01, __builtin_bit_cast(int, v[0])));
        v[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a01, __builtin_bit_cast(int, v[1])));
        v[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a23, __builtin_bit_cast(int, v[2])));
        v[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(a23, __builtin_bit_cast(int, v[3])));
    }
    // arg-max with torch.max's tie-break (first index): all-reduce the values, then the SMALLEST index among
    // the lanes that hold the maximum -- two fused butterflies instead of a (value, index) pair exchange
    template <int n>
    __device__ __forceinline__ void allreduce_argmax(float* v, int* a, int G) {
        float own[n];
#pragma unroll
        for (int k = 0; k < n; ++k) own[k] = v[k];
        butterfly<kOpMax, n>(v, G);
#pragma unroll
        for (int k = 0; k < n; ++k) a[k] = own[k] == v[k] ? a[k] : 0x7fffffff;
        butterfly<kOpMinI, n>(a, G);
    }
};


template <typename T>
__device__ __forceinline__ T* region_ptr(const Region& r, char* smem, char* wsb) {
    return reinterpret_cast<T*>((r.lds ? smem : wsb) + r.off);
}

// Kernel variant MODE: 0 ... 3 = the placement modes of DmvLayout / DepLayout; kModeShort = placement 0 (everything in LDS) with the
// short-sentence code image (vlg_dp_core.h: kSpansShort), launched for N <= kShortN.
constexpr int kModeShort = 4;
constexpr int placement_of(int mode) { return mode == kModeShort ? 0 : mode; }
constexpr int spans_of(int mode) { return mode == kModeShort ? kSpansShort : (mode != 0 ? kSpansLong : kSpansGeneral); }

template <int SR, int MODE, bool BWD>
__device__ __forceinline__ DmvCtx carve_dmv(int N, int len, char* smem, char* wsb, bool walk) {
    const DmvLayout L(N, BWD, SR == VLG_SR_MAX, placement_of(MODE), walk);
    DmvCtx c;
    c.walk = walk;
    c.Ne = len + 1;
    c.len = len;
    c.P = chart_pitch(N);
    c.C = region_ptr<float2>(L.C_in, smem, wsb);
    c.I = region_ptr<float2>(L.I_in, smem, wsb);
    c.C2 = region_ptr<float2>(L.C, smem, wsb);
    c.I2 = region_ptr<float2>(L.I, smem, wsb);
    c.S = region_ptr<float>(L.S, smem, wsb);
    c.bpS = region_ptr<unsigned char>(L.bpS, smem, wsb);
    c.bpC = region_ptr<unsigned char>(L.bpC, smem, wsb);
    c.gCc = region_ptr<float>(L.gCc, smem, wsb);
    c.gCi = region_ptr<float2>(L.gCi, smem, wsb);
    c.gI = region_ptr<float2>(L.gI, smem, wsb);
    c.decs = region_ptr<float>(L.decs, smem, wsb);
    c.gdecs = region_ptr<float>(L.gdecs, smem, wsb);
    return c;
}

template <int SR, int MODE, bool BWD>
__device__ __forceinline__ DepCtx carve_dep(int N, int len, char* smem, char* wsb) {
    const DepLayout L(N, BWD, SR == VLG_SR_MAX, placement_of(MODE));
    DepCtx c;
    c.Ne = len + 1;
    c.len = len;
    c.P = chart_pitch(N);
    c.C = region_ptr<float>(L.C, smem, wsb);
    c.I = region_ptr<float>(L.I, smem, wsb);
    c.S = region_ptr<float>(L.S, smem, wsb);
    c.bpS = region_ptr<unsigned char>(L.bpS, smem, wsb);
    c.bpC = region_ptr<unsigned char>(L.bpC, smem, wsb);
    c.gCc = region_ptr<float>(L.gCc, smem, wsb);
    c.gCi = region_ptr<float>(L.gCi, smem, wsb);
    c.gI = region_ptr<float>(L.gI, smem, wsb);
    return c;
}

// one sentence b = one workgroup: the whole DP (shared by dmv1o_kernel and, with two semirings in one grid, dmv1o_pair_kernel)
template <int SR, int MODE, bool BWD, typename In>
__device__ __forceinline__ void dmv1o_sentence(int b, const typename In::T* __restrict__ dec, const typename In::T* __restrict__ attach,
                                               const int64_t* __restrict__ lengths, int N, const float* __restrict__ glogZ,
                                               float* __restrict__ logZ, float* __restrict__ gdec, float* __restrict__ gatt,
                                               long long* __restrict__ heads, char* __restrict__ ws, size_t ws_stride, char* smem) {
    const int tid = threadIdx.x;
    const int len = (int)lengths[b];
    const size_t dec_off = (size_t)b * N * 8, att_off = (size_t)b * N * N * 2;

    if (len < 1 || len > N - 1) {   // not a sentence: NaN score, zero counts (block-uniform branch)
        if (tid == 0) logZ[b] = __uint_as_float(0x7fc00000u);
        if (BWD) {
            if (gatt) for (int i = tid; i < N * N * 2; i += kThreads) gatt[att_off + i] = 0.f;
            if (gdec) for (int i = tid; i < N * 8; i += kThreads) gdec[dec_off + i] = 0.f;
            if (heads) for (int i = tid; i < N; i += kThreads) heads[(size_t)b * N + i] = 0;
        }
        return;
    }

    char* wsb = ws + (size_t)b * ws_stride;
    const DmvCtx c = carve_dmv<SR, MODE, BWD>(N, len, smem, wsb, BWD && SR == VLG_SR_MAX);
    MergedIO<In> io;
    io.dec = dec + dec_off;
    io.attach = attach + att_off;
    io.N = N;
    io.gdec = (BWD && gdec) ? gdec + dec_off : nullptr;
    io.gatt = (BWD && gatt) ? gatt + att_off : nullptr;
    io.heads = (BWD && heads) ? heads + (size_t)b * N : nullptr;
    DevX x;
    dmv_run<SR, BWD, spans_of(MODE)>(c, io, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b, tid, kThreads, x);   // long-sentence placements: chunked long spans
}

template <int SR, int MODE, bool BWD, typename In>
__global__ __launch_bounds__(kThreads) void dmv1o_kernel(const typename In::T* __restrict__ dec,
                                                         const typename In::T* __restrict__ attach,
                                                         const int64_t* __restrict__ lengths, int N,
                                                         const float* __restrict__ glogZ, float* __restrict__ logZ,
                                                         float* __restrict__ gdec, float* __restrict__ gatt,
                                                         long long* __restrict__ heads, char* __restrict__ ws,
                                                         size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    dmv1o_sentence<SR, MODE, BWD, In>(blockIdx.x, dec, attach, lengths, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, smem);
}

// The pair lang_feat_max_tree asks for every step (joint.py:251-258) in ONE launch: blockIdx.y = 0 runs the Log semiring's
// inside-outside pass of sentence blockIdx.x (marginals), blockIdx.y = 1 the Max semiring's inside pass + back-pointer walk of the
// same sentence (best score, heads, optionally the tree counts).  Both are everything-in-LDS placements (mode 0) whose footprints
// share a CU, so the two workgroups of a sentence run side by side as they did on two streams -- without the two cross-queue
// dependencies (fork after the potentials' producer, join before the first consumer: ~15 us of a 110 us pair at B = 256, L = 40).
template <typename In, int MODE>   // MODE: 0 or kModeShort
__global__ __launch_bounds__(kThreads) void dmv1o_pair_kernel(const typename In::T* __restrict__ dec, const typename In::T* __restrict__ attach,
                                                              const int64_t* __restrict__ lengths, int N, float* __restrict__ logZ,
                                                              float* __restrict__ gdec_log, float* __restrict__ gatt_log,
                                                              float* __restrict__ best, float* __restrict__ gdec_max,
                                                              float* __restrict__ gatt_max, long long* __restrict__ heads) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.y == 0)
        dmv1o_sentence<VLG_SR_LOG, MODE, true, In>(blockIdx.x, dec, attach, lengths, N, nullptr, logZ, gdec_log, gatt_log, nullptr, nullptr, 0, smem);
    else
        dmv1o_sentence<VLG_SR_MAX, MODE, true, In>(blockIdx.x, dec, attach, lengths, N, nullptr, best, gdec_max, gatt_max, heads, nullptr, 0, smem);
}

// The same DP fed from the scorer's rule tables (RuleIO): no gathered [B,L,L,2,2]
// tensor, no masks, no merged copies in HBM.  N = L + 1 positions; gradients return in rule space.
template <int SR, int MODE, bool BWD, typename In>
__global__ __launch_bounds__(kThreads) void dmv1o_rules_kernel(
    const typename In::T* __restrict__ rule, const typename In::T* __restrict__ dec,
    const typename In::T* __restrict__ root, int root_stride, const int64_t* __restrict__ token,
    const uint8_t* __restrict__ head_mask, const int64_t* __restrict__ lengths, int Lw, int T, float fill,
    const float* __restrict__ glogZ, float* __restrict__ logZ, float* __restrict__ g_rule, float* __restrict__ g_dec,
    float* __restrict__ g_root, long long* __restrict__ heads, char* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, N = Lw + 1;
    const int len = (int)lengths[b];
    // a token id outside [0, T) (a pad / unk id) would index the rule tables -- and scatter the counts -- out of bounds:
    // such a sentence is "not a sentence", like an out-of-range length (block-uniform test)
    bool bad_tok = false;
    if (len >= 1 && len <= Lw)
        for (int i = tid; i < len; i += kThreads) {
            const long long tk = token[(size_t)b * Lw + i];
            bad_tok |= tk < 0 || tk >= T;
        }
    if (__syncthreads_or(bad_tok) || len < 1 || len > Lw) {   // outputs were zero-filled by the launcher
        if (tid == 0) logZ[b] = __uint_as_float(0x7fc00000u);
        if (BWD && heads) for (int i = tid; i < N; i += kThreads) heads[(size_t)b * N + i] = 0;
        return;
    }
    char* wsb = ws + (size_t)b * ws_stride;
    const DmvCtx c = carve_dmv<SR, MODE, BWD>(N, len, smem, wsb, BWD && SR == VLG_SR_MAX);
    RuleIO<In> io;
    io.rule = rule + (size_t)b * Lw * T * 4;
    io.dec = dec + (size_t)b * Lw * 8;
    io.root = root + (size_t)b * root_stride;
    io.token = reinterpret_cast<const long long*>(token) + (size_t)b * Lw;
    io.head_mask = head_mask ? head_mask + (size_t)b * Lw : nullptr;
    io.L = Lw;
    io.T = T;
    io.fill = fill;
    io.g_rule = (BWD && g_rule) ? g_rule + (size_t)b * Lw * T * 4 : nullptr;
    io.g_dec = (BWD && g_dec) ? g_dec + (size_t)b * Lw * 8 : nullptr;
    io.g_root = (BWD && g_root) ? g_root + (size_t)b * T : nullptr;
    io.heads = (BWD && heads) ? heads + (size_t)b * N : nullptr;
    DevX x;
    dmv_run<SR, BWD, spans_of(MODE)>(c, io, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b, tid, kThreads, x);   // long-sentence placements: chunked long spans
}

template <int SR, int MODE, bool BWD, typename In>
__global__ __launch_bounds__(kThreads) void deptree_kernel(const typename In::T* __restrict__ arc,
                                                           const int64_t* __restrict__ lengths, int N,
                                                           const float* __restrict__ glogZ, float* __restrict__ logZ,
                                                           float* __restrict__ garc, long long* __restrict__ heads,
                                                           char* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = lengths ? (int)lengths[b] : N - 1;   // lengths=None -> N-1 (deptree.py:151-152)
    const size_t arc_off = (size_t)b * N * N;
    if (len < 1 || len > N - 1) {
        if (tid == 0) logZ[b] = __uint_as_float(0x7fc00000u);
        if (BWD) {
            if (garc) for (int i = tid; i < N * N; i += kThreads) garc[arc_off + i] = 0.f;
            if (heads) for (int i = tid; i < N; i += kThreads) heads[(size_t)b * N + i] = 0;
        }
        return;
    }
    char* wsb = ws + (size_t)b * ws_stride;
    const DepCtx c = carve_dep<SR, MODE, BWD>(N, len, smem, wsb);
    DevX x;
    dep_run<SR, BWD, In, (MODE == kModeShort ? kSpansShort : kSpansGeneral)>(c, arc + arc_off, N, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b,
                         (BWD && garc) ? garc + arc_off : nullptr, (BWD && heads) ? heads + (size_t)b * N : nullptr, tid,
                         kThreads, x);
}


// ---- launch plumbing --------------------------------------------------------------------------------
struct DmvArgs {
    const void *dec, *attach;
    const int64_t* lengths;
    int B, N;
    const float* glogZ;
    float *logZ, *gdec, *gatt;
    int64_t* heads;
    void* ws;
    size_t ws_stride, lds;
    hipStream_t s;
};

struct RulesArgs {
    const void *rule, *dec, *root;
    int root_stride;
    const int64_t* token;
    const uint8_t* head_mask;
    const int64_t* lengths;
    int B, L, T;
    float fill;
    const float* glogZ;
    float *logZ, *g_rule, *g_dec, *g_root;
    int64_t* heads;
    void* ws;
    size_t ws_stride, lds;
    hipStream_t s;
};

struct DepArgs {
    const void* arc;
    const int64_t* lengths;
    int B, N;
    const float* glogZ;
    float *logZ, *garc;
    int64_t* heads;
    void* ws;
    size_t ws_stride, lds;
    hipStream_t s;
};

// One entry point per (family, semiring, input type), each defined by its own compilation of vlg_dp_inst.hip.
// family: 0 = DMV1o on merged potentials, 1 = DMV1o on rule tables, 2 = DepTree; sr: VLG_SR_*; in: VLG_F32 / VLG_BF16.
#define VLG_DP_INST_NAME_(F, S, I) dp_inst_##F##_##S##_##I
#define VLG_DP_INST_NAME(F, S, I) VLG_DP_INST_NAME_(F, S, I)
#define VLG_DP_DECLARE(F, ARGS)                                      \
    int VLG_DP_INST_NAME(F, 0, 0)(bool bwd, int mode, const ARGS& a); \
    int VLG_DP_INST_NAME(F, 0, 1)(bool bwd, int mode, const ARGS& a); \
    int VLG_DP_INST_NAME(F, 1, 0)(bool bwd, int mode, const ARGS& a); \
    int VLG_DP_INST_NAME(F, 1, 1)(bool bwd, int mode, const ARGS& a);
VLG_DP_DECLARE(0, DmvArgs)
VLG_DP_DECLARE(1, RulesArgs)
VLG_DP_DECLARE(2, DepArgs)
#undef VLG_DP_DECLARE

template <typename K>
static int prep(K kernel, size_t lds) {
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", lds,
                                              hipGetErrorString(e));
    }
    return 0;
}

template <int SR, int MODE, bool BWD, typename In>
static int launch_dmv(const DmvArgs& a) {
    auto k = dmv1o_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, a.lds)) return rc;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(kThreads), a.lds, a.s, (const typename In::T*)a.dec,
                       (const typename In::T*)a.attach, a.lengths, a.N, a.glogZ, a.logZ, a.gdec, a.gatt,
                       (long long*)a.heads, (char*)a.ws, a.ws_stride);
    return check_launch("dmv1o_kernel");
}

template <int SR, int MODE, bool BWD, typename In>
static int launch_rules(const RulesArgs& a) {
    auto k = dmv1o_rules_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, a.lds)) return rc;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(kThreads), a.lds, a.s, (const typename In::T*)a.rule,
                       (const typename In::T*)a.dec, (const typename In::T*)a.root, a.root_stride, a.token, a.head_mask,
                       a.lengths, a.L, a.T, a.fill, a.glogZ, a.logZ, a.g_rule, a.g_dec, a.g_root, (long long*)a.heads,
                       (char*)a.ws, a.ws_stride);
    return check_launch("dmv1o_rules_kernel");
}

template <int SR, int MODE, bool BWD, typename In>
static int launch_dep(const DepArgs& a) {
    auto k = deptree_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, a.lds)) return rc;
    hipLaunchKernelGGL(k, dim3(a.B), dim3(kThreads), a.lds, a.s, (const typename In::T*)a.arc, a.lengths, a.N, a.glogZ,
                       a.logZ, a.garc, (long long*)a.heads, (char*)a.ws, a.ws_stride);
    return check_launch("deptree_kernel");
}

}  // namespace vlg
