// vlg_dp.hip -- gfx950 kernels for the batched structured DP (DMV1o and DepTree, inside + outside)
// and their C-ABI entry points (declared in include/vlgae_amd.h).
//
// One workgroup per sentence.  Charts live in LDS (160 KiB per CU on CDNA4); the outside pass
// is an explicit adjoint replay of the inside loop in the SAME launch, so the only HBM traffic
// is the algorithmic minimum: potentials in, logZ and expected counts out.  Sentences longer
// than the LDS budget spill the value charts (MODE 1) or everything (MODE 2) to a caller-owned
// workspace that stays L2 / Infinity-Cache resident.
//
// Reference behaviour reproduced: src/model/torch_struct/dmv.py:19-66, deptree.py:25-76,
// helpers.py:101-157 (sum / marginals via autograd), distributions.py:253-265 (merge).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_dp_core.h"

namespace vlg {

#ifndef VLG_DP_THREADS
#define VLG_DP_THREADS 512
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

template <int OPK, int n, typename T>
__device__ __forceinline__ void butterfly(T* v, int G) {   // G is uniform over the workgroup: no divergence
    if (G > 1) butterfly_step<OPK, 1, n>(v);
    if (G > 2) butterfly_step<OPK, 2, n>(v);
    if (G > 4) butterfly_step<OPK, 4, n>(v);
    if (G > 8) butterfly_step<OPK, 8, n>(v);
    if (G > 16) butterfly_step<OPK, 16, n>(v);
    if (G > 32) butterfly_step<OPK, 32, n>(v);
}

struct DevX {
    static constexpr bool kSkipDeadWaves = true;   // the all-reduces are wave-local: a wave without spans can skip a phase
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

// self-test of the exchange primitives against __shfl_xor on block-uniform data (tests/test_gpu_parity.py)
__global__ void xlane_selftest_kernel(int* out) {
    const int lane = threadIdx.x & 63;
    int bad = 0;
#define VLG_CHK(K)                                                                    \
    {                                                                                 \
        const int v = (lane / K) * 1000 + 7;        /* uniform over aligned K-blocks */ \
        if (xlane_i<K>(v) != __shfl_xor(v, K, 64)) bad |= K;                          \
    }
    VLG_CHK(1) VLG_CHK(2) VLG_CHK(4) VLG_CHK(8) VLG_CHK(16) VLG_CHK(32)
#undef VLG_CHK
    float m[2] = {(float)((lane * 37) % 64), -(float)lane};
    int am[2] = {lane, lane};
    DevX x;
    x.allreduce_argmax<2>(m, am, 64);
    if (m[0] != 63.f || m[1] != 0.f || am[1] != 0) bad |= 128;
    float s[1] = {1.0f};
    x.allreduce_sum<1>(s, 16);
    if (s[0] != 16.f) bad |= 256;
    float s6[6], m6[6];
    for (int k = 0; k < 6; ++k) { s6[k] = (float)(k + 1); m6[k] = (float)((lane * (k + 3)) % 61); }
    x.allreduce_sum<6>(s6, 64);
    x.allreduce_max<6>(m6, 64);
    for (int k = 0; k < 6; ++k)
        if (s6[k] != 64.f * (k + 1) || m6[k] != 60.f) bad |= 512;
    float s4[4] = {1.f, 2.f, 3.f, (float)lane};
    x.allreduce_sum<4>(s4, 32);
    if (s4[0] != 32.f || s4[2] != 96.f || s4[3] != (lane < 32 ? 496.f : 1520.f)) bad |= 1024;
    atomicOr(out, bad);
}

template <typename T>
__device__ __forceinline__ T* region_ptr(const Region& r, char* smem, char* wsb) {
    return reinterpret_cast<T*>((r.lds ? smem : wsb) + r.off);
}

template <int SR, int MODE, bool BWD>
__device__ __forceinline__ DmvCtx carve_dmv(int N, int len, char* smem, char* wsb, bool walk) {
    const DmvLayout L(N, BWD, SR == VLG_SR_MAX, MODE, walk);
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

template <int SR, int MODE, bool BWD, typename In>
__global__ __launch_bounds__(kThreads) void dmv1o_kernel(const typename In::T* __restrict__ dec,
                                                         const typename In::T* __restrict__ attach,
                                                         const int64_t* __restrict__ lengths, int N,
                                                         const float* __restrict__ glogZ, float* __restrict__ logZ,
                                                         float* __restrict__ gdec, float* __restrict__ gatt,
                                                         long long* __restrict__ heads, char* __restrict__ ws,
                                                         size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
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
    const DmvCtx c = carve_dmv<SR, MODE, BWD>(N, len, smem, wsb, BWD && SR == VLG_SR_MAX && gdec == nullptr);
    MergedIO<In> io;
    io.dec = dec + dec_off;
    io.attach = attach + att_off;
    io.N = N;
    io.gdec = (BWD && gdec) ? gdec + dec_off : nullptr;
    io.gatt = (BWD && gatt) ? gatt + att_off : nullptr;
    io.heads = (BWD && heads) ? heads + (size_t)b * N : nullptr;
    DevX x;
    dmv_run<SR, BWD, (MODE != 0)>(c, io, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b, tid, kThreads, x);   // long-sentence placements: chunked long spans
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
    const DmvCtx c = carve_dmv<SR, MODE, BWD>(N, len, smem, wsb, BWD && SR == VLG_SR_MAX && g_dec == nullptr);
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
    dmv_run<SR, BWD, (MODE != 0)>(c, io, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b, tid, kThreads, x);   // long-sentence placements: chunked long spans
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
    const DepLayout L(N, BWD, SR == VLG_SR_MAX, MODE);
    char* wsb = ws + (size_t)b * ws_stride;
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
    DevX x;
    dep_run<SR, BWD, In>(c, arc + arc_off, N, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b,
                         (BWD && garc) ? garc + arc_off : nullptr, (BWD && heads) ? heads + (size_t)b * N : nullptr, tid,
                         kThreads, x);
}

// ---- DMV1o.merge (distributions.py:253-265): root-augmented potentials, always fp32 out ----------
template <typename In>
__global__ void merge_kernel(const typename In::T* __restrict__ dec, const typename In::T* __restrict__ attach,
                             const typename In::T* __restrict__ root, int B, int Lw, float one, float zero,
                             float* __restrict__ dec_wroot, float* __restrict__ attach_wroot) {
    const int N = Lw + 1;
    const size_t n_att = (size_t)B * N * N * 2, n_dec = (size_t)B * N * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n_att + n_dec;
         idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < n_att) {
            const int v = idx & 1;
            size_t t = idx >> 1;
            const int ch = t % N; t /= N;
            const int h = t % N;
            const size_t b = t / N;
            float val = zero;
            if (ch >= 1) {
                if (h == 0) { if (v == 1) val = In::ld(root, b * Lw + (ch - 1)); }        // attach[:,0,1:,NOCHILD] = root
                else val = In::ld(attach, ((b * Lw + (h - 1)) * Lw + (ch - 1)) * 2 + v);    // attach[:,1:,1:,:] = attach
            }
            attach_wroot[idx] = val;
        } else {
            const size_t j = idx - n_att;
            const int k = j & 7;
            const size_t t = j >> 3;
            const int h = t % N;
            const size_t b = t / N;
            float val;
            if (h == 0) val = (k >> 2) == 1 ? one : zero;                                   // dec[:,0,RIGHT,:,:] = one
            else val = In::ld(dec, (b * Lw + (h - 1)) * 8 + k);                             // dec[:,1:] = dec
            dec_wroot[j] = val;
        }
    }
}

// ---- batch sum of the expected counts: out[m] = sum_b counts[b][m] over the concatenation [grad_dec | grad_attach].
// This is the marginal-loss gradient of position-tied parameters -- the quantity the data-parallel all-reduce
// carries in bench.py (the reference's DDP gradient sum, config/trainer/train.yaml:27-29).  One block = 64 columns x
// 16 row groups; fixed summation order (no atomics), coalesced 256-byte rows per wavefront.
__global__ __launch_bounds__(1024) void count_sum_kernel(const float* __restrict__ gdec, const float* __restrict__ gatt,
                                                         int B, int Md, int Ma, float* __restrict__ out) {
    __shared__ float part[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < Md + Ma) {
        const float* src = col < Md ? gdec + col : gatt + (col - Md);
        const size_t pitch = col < Md ? Md : Ma;
#pragma unroll 4
        for (int b = rg; b < B; b += 16) acc += src[(size_t)b * pitch];
    }
    part[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rg == 0 && col < Md + Ma) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
        out[col] = t;
    }
}

// Chain rule of the partition function on the API path: the fused launch keeps unit-upstream counts, autograd later hands
// over d loss / d logZ[b]; this scales both count tensors by it and writes them in the potentials' storage type -- one
// launch instead of two multiplies and two casts (the API path is host-bound).  g_stride 0: one scalar for
// the whole batch (the expanded gradient of a `.sum()`).
template <typename Out>
__global__ __launch_bounds__(256) void scale_counts_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           const float* __restrict__ g, int g_stride, int na, int nb,
                                                           Out* __restrict__ oa, Out* __restrict__ ob) {
    const int s = blockIdx.y;
    const float gs = g[(size_t)s * g_stride];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < na) oa[(size_t)s * na + i] = (Out)(a[(size_t)s * na + i] * gs);
    else if (i - na < nb) ob[(size_t)s * nb + (i - na)] = (Out)(b[(size_t)s * nb + (i - na)] * gs);
}

// ---- launch plumbing --------------------------------------------------------------------------------
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
static int launch_dmv(const void* dec, const void* attach, const int64_t* lengths, int B, int N, const float* glogZ,
                      float* logZ, float* gdec, float* gatt, int64_t* heads, void* ws, size_t ws_stride,
                      size_t lds, hipStream_t stream) {
    auto k = dmv1o_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), lds, stream, (const typename In::T*)dec,
                       (const typename In::T*)attach, lengths, N, glogZ, logZ, gdec, gatt, (long long*)heads, (char*)ws,
                       ws_stride);
    return check_launch("dmv1o_kernel");
}

template <int SR, bool BWD, typename In>
static int dispatch_dmv_mode(int mode, const void* dec, const void* attach, const int64_t* lengths, int B, int N,
                             const float* glogZ, float* logZ, float* gdec, float* gatt, int64_t* heads, void* ws,
                             size_t ws_stride, size_t lds, hipStream_t s) {
    switch (mode) {
        case 0: return launch_dmv<SR, 0, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s);
        case 1: return launch_dmv<SR, 1, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s);
        case 2: return launch_dmv<SR, 2, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s);
        default: return launch_dmv<SR, 3, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s);
    }
}

template <bool BWD>
static int run_dmv(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                   int semiring, const float* glogZ, float* logZ, float* gdec, float* gatt, int64_t* heads, void* ws,
                   size_t ws_bytes, void* stream) {
    if (B < 0 || N < 2) return set_error(VLG_ERR_SHAPE, "dmv1o: need B >= 0 and N >= 2 (got B=%d N=%d)", B, N);
    if (N > 255) return set_error(VLG_ERR_SHAPE, "dmv1o: N=%d exceeds the supported maximum of 255", N);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "dmv1o: in_dtype %d", in_dtype);
    if (semiring != VLG_SR_LOG && semiring != VLG_SR_MAX) return set_error(VLG_ERR_ARG, "dmv1o: semiring %d", semiring);
    if (!dec || !attach || !lengths || !logZ || (BWD && !heads && !gatt))   // gdec may be null: attach counts only
        if (B > 0) return set_error(VLG_ERR_ARG, "dmv1o: null buffer");
    if (B == 0) return 0;
    const bool walk = BWD && semiring == VLG_SR_MAX && !gdec;   // tree only: lean layout, back-pointer walk (the kernel derives the same flag)
    const int mode = pick_mode<DmvLayout>(N, BWD, semiring == VLG_SR_MAX, kLdsBudget, walk);
    const DmvLayout L(N, BWD, semiring == VLG_SR_MAX, mode, walk);
    const size_t ws_stride = L.ws_bytes, lds = L.lds_bytes;
    if (ws_stride * (size_t)B > ws_bytes || (ws_stride && !ws))
        return set_error(VLG_ERR_WORKSPACE, "dmv1o: N=%d needs a %zu-byte workspace (got %zu); see vlg_workspace_bytes",
                         N, ws_stride * (size_t)B, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
#ifdef VLG_DP_HEADLINE_ONLY   // tools/ A/B builds: only the Log / all-in-LDS / bf16 instantiation (compiles in seconds)
    if (semiring != VLG_SR_LOG || mode > 1 || in_dtype != VLG_BF16) return set_error(VLG_ERR_ARG, "headline-only build");
    if (mode == 1) return launch_dmv<VLG_SR_LOG, 1, BWD, BF16In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s);
    return launch_dmv<VLG_SR_LOG, 0, BWD, BF16In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s);
#else
#define VLG_GO(SRV, INV) \
    return dispatch_dmv_mode<SRV, BWD, INV>(mode, dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, s)
    if (semiring == VLG_SR_LOG) {
        if (in_dtype == VLG_F32) VLG_GO(VLG_SR_LOG, F32In);
        VLG_GO(VLG_SR_LOG, BF16In);
    }
    if (in_dtype == VLG_F32) VLG_GO(VLG_SR_MAX, F32In);
    VLG_GO(VLG_SR_MAX, BF16In);
#undef VLG_GO
#endif
}

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

template <int SR, bool BWD, typename In>
static int dispatch_rules_mode(int mode, const RulesArgs& a) {
    switch (mode) {
        case 0: return launch_rules<SR, 0, BWD, In>(a);
        case 1: return launch_rules<SR, 1, BWD, In>(a);
        case 2: return launch_rules<SR, 2, BWD, In>(a);
        default: return launch_rules<SR, 3, BWD, In>(a);
    }
}

template <int SR, int MODE, bool BWD, typename In>
static int launch_dep(const void* arc, const int64_t* lengths, int B, int N, const float* glogZ, float* logZ,
                      float* garc, int64_t* heads, void* ws, size_t ws_stride, size_t lds, hipStream_t stream) {
    auto k = deptree_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), lds, stream, (const typename In::T*)arc, lengths, N, glogZ, logZ,
                       garc, (long long*)heads, (char*)ws, ws_stride);
    return check_launch("deptree_kernel");
}

template <int SR, bool BWD, typename In>
static int dispatch_dep_mode(int mode, const void* arc, const int64_t* lengths, int B, int N, const float* glogZ,
                             float* logZ, float* garc, int64_t* heads, void* ws, size_t ws_stride, size_t lds,
                             hipStream_t s) {
    switch (mode) {
        case 0: return launch_dep<SR, 0, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, heads, ws, ws_stride, lds, s);
        case 1: return launch_dep<SR, 1, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, heads, ws, ws_stride, lds, s);
        case 2: return launch_dep<SR, 2, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, heads, ws, ws_stride, lds, s);
        default: return launch_dep<SR, 3, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, heads, ws, ws_stride, lds, s);
    }
}

template <bool BWD>
static int run_dep(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring,
                   const float* glogZ, float* logZ, float* garc, int64_t* heads, void* ws, size_t ws_bytes,
                   void* stream) {
    if (B < 0 || N < 2) return set_error(VLG_ERR_SHAPE, "deptree: need B >= 0 and N >= 2 (got B=%d N=%d)", B, N);
    if (N > 255) return set_error(VLG_ERR_SHAPE, "deptree: N=%d exceeds the supported maximum of 255", N);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "deptree: in_dtype %d", in_dtype);
    if (semiring != VLG_SR_LOG && semiring != VLG_SR_MAX) return set_error(VLG_ERR_ARG, "deptree: semiring %d", semiring);
    if (!arc || !logZ || (BWD && !garc && !heads))
        if (B > 0) return set_error(VLG_ERR_ARG, "deptree: null buffer");
    if (B == 0) return 0;
    const int mode = pick_mode<DepLayout>(N, BWD, semiring == VLG_SR_MAX, kLdsBudget);
    const DepLayout L(N, BWD, semiring == VLG_SR_MAX, mode);
    const size_t ws_stride = L.ws_bytes, lds = L.lds_bytes;
    if (ws_stride * (size_t)B > ws_bytes || (ws_stride && !ws))
        return set_error(VLG_ERR_WORKSPACE, "deptree: N=%d needs a %zu-byte workspace (got %zu)", N,
                         ws_stride * (size_t)B, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
#ifdef VLG_DP_HEADLINE_ONLY
    (void)s; (void)lds;
    return set_error(VLG_ERR_ARG, "headline-only build");
#else
#define VLG_GO(SRV, INV) \
    return dispatch_dep_mode<SRV, BWD, INV>(mode, arc, lengths, B, N, glogZ, logZ, garc, heads, ws, ws_stride, lds, s)
    if (semiring == VLG_SR_LOG) {
        if (in_dtype == VLG_F32) VLG_GO(VLG_SR_LOG, F32In);
        VLG_GO(VLG_SR_LOG, BF16In);
    }
    if (in_dtype == VLG_F32) VLG_GO(VLG_SR_MAX, F32In);
    VLG_GO(VLG_SR_MAX, BF16In);
#undef VLG_GO
#endif
}

}  // namespace vlg

// =====================================================================================================
// C ABI (include/vlgae_amd.h)
// =====================================================================================================
extern "C" {

int vlg_dmv1o_inside(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                     int semiring, float* logZ, void* ws, size_t ws_bytes, void* stream) {
    return vlg::run_dmv<false>(dec, attach, lengths, B, N, in_dtype, semiring, nullptr, logZ, nullptr, nullptr, nullptr,
                               ws, ws_bytes, stream);
}

int vlg_dmv1o_inside_outside(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                             int semiring, const float* grad_logZ, float* logZ, float* grad_dec, float* grad_attach,
                             void* ws, size_t ws_bytes, void* stream) {
    return vlg::run_dmv<true>(dec, attach, lengths, B, N, in_dtype, semiring, grad_logZ, logZ, grad_dec, grad_attach,
                              nullptr, ws, ws_bytes, stream);
}

int vlg_deptree_inside(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring, float* logZ,
                       void* ws, size_t ws_bytes, void* stream) {
    return vlg::run_dep<false>(arc, lengths, B, N, in_dtype, semiring, nullptr, logZ, nullptr, nullptr, ws, ws_bytes, stream);
}

int vlg_deptree_inside_outside(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring,
                               const float* grad_logZ, float* logZ, float* grad_arc, void* ws, size_t ws_bytes,
                               void* stream) {
    return vlg::run_dep<true>(arc, lengths, B, N, in_dtype, semiring, grad_logZ, logZ, grad_arc, nullptr, ws, ws_bytes, stream);
}

int vlg_dmv1o_decode(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                     float* best_score, int64_t* heads, void* ws, size_t ws_bytes, void* stream) {
    if (B > 0 && !heads) return vlg::set_error(VLG_ERR_ARG, "dmv1o_decode: null heads");
    return vlg::run_dmv<true>(dec, attach, lengths, B, N, in_dtype, VLG_SR_MAX, nullptr, best_score, nullptr, nullptr,
                              heads, ws, ws_bytes, stream);
}

int vlg_deptree_decode(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, float* best_score,
                       int64_t* heads, void* ws, size_t ws_bytes, void* stream) {
    if (B > 0 && !heads) return vlg::set_error(VLG_ERR_ARG, "deptree_decode: null heads");
    return vlg::run_dep<true>(arc, lengths, B, N, in_dtype, VLG_SR_MAX, nullptr, best_score, nullptr, heads, ws,
                              ws_bytes, stream);
}

int vlg_dmv1o_rules(const void* attach_rule, const void* dec, const void* root_rule, int root_per_sentence,
                    const int64_t* token, const uint8_t* head_mask, const int64_t* lengths, int B, int L, int T,
                    int in_dtype, int semiring, float mask_fill, const float* grad_logZ, float* logZ, float* grad_rule,
                    float* grad_dec, float* grad_root, int64_t* heads, void* ws, size_t ws_bytes, void* stream) {
    using namespace vlg;
    const int N = L + 1;
    if (B < 0 || L < 1 || T < 1) return set_error(VLG_ERR_SHAPE, "dmv1o_rules: bad shape B=%d L=%d T=%d", B, L, T);
    if (N > 255) return set_error(VLG_ERR_SHAPE, "dmv1o_rules: L=%d exceeds the supported maximum of 254", L);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "dmv1o_rules: in_dtype %d", in_dtype);
    if (semiring != VLG_SR_LOG && semiring != VLG_SR_MAX) return set_error(VLG_ERR_ARG, "dmv1o_rules: semiring %d", semiring);
    if (B == 0) return 0;
    if (!attach_rule || !dec || !root_rule || !token || !lengths || !logZ) return set_error(VLG_ERR_ARG, "dmv1o_rules: null buffer");
    if (heads && semiring != VLG_SR_MAX)
        return set_error(VLG_ERR_ARG, "dmv1o_rules: a head vector is the Viterbi tree -- pass semiring = VLG_SEMIRING_MAX with heads");
    const bool bwd = grad_rule || grad_dec || grad_root || heads;
    if (bwd && !heads && !(grad_rule && grad_dec && grad_root))
        return set_error(VLG_ERR_ARG, "dmv1o_rules: pass all three gradient buffers (or only heads)");
    const bool is_max = semiring == VLG_SR_MAX;
    const bool walk = bwd && is_max && !grad_dec;
    const int mode = pick_mode<DmvLayout>(N, bwd, is_max, kLdsBudget, walk);
    const DmvLayout Lay(N, bwd, is_max, mode, walk);
    if (Lay.ws_bytes * (size_t)B > ws_bytes || (Lay.ws_bytes && !ws))
        return set_error(VLG_ERR_WORKSPACE, "dmv1o_rules: L=%d needs a %zu-byte workspace (got %zu)", L, Lay.ws_bytes * (size_t)B, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
    if (grad_rule) {   // rule-space counts are accumulated with atomics: start from zero
        hipError_t e = hipMemsetAsync(grad_rule, 0, sizeof(float) * (size_t)B * L * T * 4, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_root, 0, sizeof(float) * (size_t)B * T, s);
        if (e == hipSuccess) e = hipMemsetAsync(grad_dec, 0, sizeof(float) * (size_t)B * L * 8, s);
        if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    RulesArgs a{attach_rule, dec, root_rule, root_per_sentence ? T : 0, token, head_mask, lengths, B, L, T, mask_fill,
                grad_logZ, logZ, grad_rule, grad_dec, grad_root, heads, ws, Lay.ws_bytes, Lay.lds_bytes, s};
#ifdef VLG_DP_HEADLINE_ONLY
    (void)a;
    return set_error(VLG_ERR_ARG, "headline-only build");
#else
#define VLG_GO(SRV, INV) return bwd ? dispatch_rules_mode<SRV, true, INV>(mode, a) : dispatch_rules_mode<SRV, false, INV>(mode, a)
    if (!is_max) {
        if (in_dtype == VLG_F32) VLG_GO(VLG_SR_LOG, F32In);
        VLG_GO(VLG_SR_LOG, BF16In);
    }
    if (in_dtype == VLG_F32) VLG_GO(VLG_SR_MAX, F32In);
    VLG_GO(VLG_SR_MAX, BF16In);
#undef VLG_GO
#endif
}

size_t vlg_workspace_bytes(int op, int B, int N, int semiring) {
    if (B <= 0 || N < 2) return 0;
    const bool is_max = semiring == VLG_SR_MAX;
    using namespace vlg;
    switch (op) {
        case VLG_OP_DMV1O_INSIDE:
            return DmvLayout(N, false, is_max, pick_mode<DmvLayout>(N, false, is_max, kLdsBudget)).ws_bytes * B;
        case VLG_OP_DMV1O_INSIDE_OUTSIDE: {   // covers both the replay layout and (Max, tree only) the leaner walk layout
            const size_t full = DmvLayout(N, true, is_max, pick_mode<DmvLayout>(N, true, is_max, kLdsBudget)).ws_bytes;
            const size_t lean = is_max ? DmvLayout(N, true, true, pick_mode<DmvLayout>(N, true, true, kLdsBudget, true), true).ws_bytes : 0;
            return (full > lean ? full : lean) * B;
        }
        case VLG_OP_DEPTREE_INSIDE:
            return DepLayout(N, false, is_max, pick_mode<DepLayout>(N, false, is_max, kLdsBudget)).ws_bytes * B;
        case VLG_OP_DEPTREE_INSIDE_OUTSIDE:
            return DepLayout(N, true, is_max, pick_mode<DepLayout>(N, true, is_max, kLdsBudget)).ws_bytes * B;
        default: return 0;
    }
}

/* Debug aid used by the GPU tests: checks the DPP / swizzle lane-exchange primitives on this device.
 * Returns 0 when they behave as the kernels assume; `scratch` is one device int. */
int vlg_selftest_xlane(int* scratch, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(int), s);
    if (e != hipSuccess) return vlg::set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(vlg::xlane_selftest_kernel, dim3(1), dim3(128), 0, s, scratch);
    return vlg::check_launch("xlane_selftest_kernel");
}

int vlg_dmv1o_merge(const void* dec, const void* attach, const void* root, int B, int L, int in_dtype, float one,
                    float zero, float* dec_wroot, float* attach_wroot, void* stream) {
    if (B < 0 || L < 1) return vlg::set_error(VLG_ERR_SHAPE, "merge: need B >= 0 and L >= 1 (got B=%d L=%d)", B, L);
    if (B == 0) return 0;
    if (!dec || !attach || !root || !dec_wroot || !attach_wroot) return vlg::set_error(VLG_ERR_ARG, "merge: null buffer");
    const size_t total = (size_t)B * (L + 1) * ((size_t)(L + 1) * 2 + 8);
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 4096 ? (total + threads - 1) / threads : 4096);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == VLG_F32)
        hipLaunchKernelGGL(vlg::merge_kernel<vlg::F32In>, dim3(blocks), dim3(threads), 0, s, (const float*)dec,
                           (const float*)attach, (const float*)root, B, L, one, zero, dec_wroot, attach_wroot);
    else if (in_dtype == VLG_BF16)
        hipLaunchKernelGGL(vlg::merge_kernel<vlg::BF16In>, dim3(blocks), dim3(threads), 0, s, (const uint16_t*)dec,
                           (const uint16_t*)attach, (const uint16_t*)root, B, L, one, zero, dec_wroot, attach_wroot);
    else
        return vlg::set_error(VLG_ERR_DTYPE, "merge: in_dtype %d", in_dtype);
    return vlg::check_launch("merge_kernel");
}

int vlg_dmv1o_count_sum(const float* grad_dec, const float* grad_attach, int B, int N, float* out, void* stream) {
    if (B < 0 || N < 2) return vlg::set_error(VLG_ERR_SHAPE, "count_sum: need B >= 0 and N >= 2 (got B=%d N=%d)", B, N);
    if (!grad_dec || !grad_attach || !out) return vlg::set_error(VLG_ERR_ARG, "count_sum: null buffer");
    const int Md = N * 8, Ma = N * N * 2;
    hipLaunchKernelGGL(vlg::count_sum_kernel, dim3((Md + Ma + 63) / 64), dim3(1024), 0, (hipStream_t)stream, grad_dec,
                       grad_attach, B, Md, Ma, out);
    return vlg::check_launch("count_sum_kernel");
}

int vlg_scale_counts(const float* counts_a, const float* counts_b, const float* g, int g_stride, int B, int n_a, int n_b,
                     int out_dtype, void* out_a, void* out_b, void* stream) {
    if (B < 0 || n_a < 0 || n_b < 0 || (g_stride != 0 && g_stride != 1))
        return vlg::set_error(VLG_ERR_SHAPE, "scale_counts: bad sizes (B=%d n_a=%d n_b=%d g_stride=%d)", B, n_a, n_b, g_stride);
    if (!g || (n_a && (!counts_a || !out_a)) || (n_b && (!counts_b || !out_b))) return vlg::set_error(VLG_ERR_ARG, "scale_counts: null buffer");
    if (out_dtype != VLG_F32 && out_dtype != VLG_BF16) return vlg::set_error(VLG_ERR_DTYPE, "scale_counts: out_dtype %d", out_dtype);
    if (B == 0 || n_a + n_b == 0) return 0;
    const dim3 grid((n_a + n_b + 255) / 256, B);
    if (out_dtype == VLG_F32)
        hipLaunchKernelGGL(vlg::scale_counts_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, counts_a, counts_b, g, g_stride, n_a,
                           n_b, (float*)out_a, (float*)out_b);
    else
        hipLaunchKernelGGL(vlg::scale_counts_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, counts_a, counts_b, g,
                           g_stride, n_a, n_b, (__bf16*)out_a, (__bf16*)out_b);
    return vlg::check_launch("scale_counts_kernel");
}

}  // extern "C"
