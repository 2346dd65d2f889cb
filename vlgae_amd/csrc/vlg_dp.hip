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

constexpr int kThreads = 256;               // 4 wave64 = one wave per SIMD of the CU
constexpr size_t kLdsBudget = 160 * 1024;   // CDNA4 LDS per CU / per workgroup

// MODE 0: everything in LDS.  MODE 1: value charts (C, I, S, back-pointers) in the global workspace,
// adjoints + dec staging in LDS.  MODE 2: everything in the global workspace.
__host__ inline int pick_mode(const DmvLayout& L) {
    if (L.total <= kLdsBudget) return 0;
    if (L.total - L.value_end <= kLdsBudget) return 1;
    return 2;
}
__host__ inline size_t lds_bytes(const DmvLayout& L, int mode) {
    return mode == 0 ? L.total : mode == 1 ? L.total - L.value_end : 0;
}
__host__ inline size_t ws_bytes_per_sentence(const DmvLayout& L, int mode) {
    return mode == 0 ? 0 : mode == 1 ? L.value_end : L.total;
}

template <int SR, int MODE, bool BWD, typename In>
__global__ __launch_bounds__(kThreads) void dmv1o_kernel(const typename In::T* __restrict__ dec,
                                                         const typename In::T* __restrict__ attach,
                                                         const int64_t* __restrict__ lengths, int N,
                                                         const float* __restrict__ glogZ, float* __restrict__ logZ,
                                                         float* __restrict__ gdec, float* __restrict__ gatt,
                                                         char* __restrict__ ws, size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = (int)lengths[b];
    const size_t dec_off = (size_t)b * N * 8, att_off = (size_t)b * N * N * 2;

    if (len < 1 || len > N - 1) {   // not a sentence: NaN score, zero counts (block-uniform branch)
        if (tid == 0) logZ[b] = __uint_as_float(0x7fc00000u);
        if (BWD) {
            for (int i = tid; i < N * N * 2; i += kThreads) gatt[att_off + i] = 0.f;
            for (int i = tid; i < N * 8; i += kThreads) gdec[dec_off + i] = 0.f;
        }
        return;
    }

    const DmvLayout L(N, BWD, SR == VLG_SR_MAX);
    char* wsb = ws + (size_t)b * ws_stride;
    auto vptr = [&](size_t off) -> char* { return MODE == 0 ? smem + off : wsb + off; };            // value charts
    auto aptr = [&](size_t off) -> char* {                                                           // adjoints, staging
        return MODE == 2 ? wsb + off : smem + (off - (MODE == 1 ? L.value_end : 0));
    };
    DmvCtx c;
    c.Ne = len + 1;
    c.len = len;
    c.P = chart_pitch(N);
    c.C = reinterpret_cast<float2*>(vptr(L.C));
    c.I = reinterpret_cast<float2*>(vptr(L.I));
    c.S = reinterpret_cast<float*>(vptr(L.S));
    c.bpS = reinterpret_cast<unsigned char*>(vptr(L.bpS));
    c.bpC = reinterpret_cast<unsigned char*>(vptr(L.bpC));
    c.gC = reinterpret_cast<float2*>(aptr(L.gC));
    c.gI = reinterpret_cast<float2*>(aptr(L.gI));
    c.decs = reinterpret_cast<float*>(aptr(L.decs));
    c.gdecs = reinterpret_cast<float*>(aptr(L.gdecs));
    dmv_run<SR, BWD, In>(c, dec + dec_off, attach + att_off, N, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b,
                         BWD ? gdec + dec_off : nullptr, BWD ? gatt + att_off : nullptr, tid, kThreads,
                         [] { __syncthreads(); });
}

// ---- DepTree --------------------------------------------------------------------------------------
__host__ inline int pick_mode(const DepLayout& L) {
    if (L.total <= kLdsBudget) return 0;
    if (L.total - L.value_end <= kLdsBudget) return 1;
    return 2;
}
__host__ inline size_t lds_bytes(const DepLayout& L, int mode) {
    return mode == 0 ? L.total : mode == 1 ? L.total - L.value_end : 0;
}
__host__ inline size_t ws_bytes_per_sentence(const DepLayout& L, int mode) {
    return mode == 0 ? 0 : mode == 1 ? L.value_end : L.total;
}

template <int SR, int MODE, bool BWD, typename In>
__global__ __launch_bounds__(kThreads) void deptree_kernel(const typename In::T* __restrict__ arc,
                                                           const int64_t* __restrict__ lengths, int N,
                                                           const float* __restrict__ glogZ, float* __restrict__ logZ,
                                                           float* __restrict__ garc, char* __restrict__ ws,
                                                           size_t ws_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = lengths ? (int)lengths[b] : N - 1;   // lengths=None -> N-1 (deptree.py:151-152)
    const size_t arc_off = (size_t)b * N * N;
    if (len < 1 || len > N - 1) {
        if (tid == 0) logZ[b] = __uint_as_float(0x7fc00000u);
        if (BWD)
            for (int i = tid; i < N * N; i += kThreads) garc[arc_off + i] = 0.f;
        return;
    }
    const DepLayout L(N, BWD, SR == VLG_SR_MAX);
    char* wsb = ws + (size_t)b * ws_stride;
    auto vptr = [&](size_t off) -> char* { return MODE == 0 ? smem + off : wsb + off; };
    auto aptr = [&](size_t off) -> char* {
        return MODE == 2 ? wsb + off : smem + (off - (MODE == 1 ? L.value_end : 0));
    };
    DepCtx c;
    c.Ne = len + 1;
    c.len = len;
    c.P = chart_pitch(N);
    c.C = reinterpret_cast<float*>(vptr(L.C));
    c.I = reinterpret_cast<float*>(vptr(L.I));
    c.S = reinterpret_cast<float*>(vptr(L.S));
    c.bpS = reinterpret_cast<unsigned char*>(vptr(L.bpS));
    c.bpC = reinterpret_cast<unsigned char*>(vptr(L.bpC));
    c.gC = reinterpret_cast<float*>(aptr(L.gC));
    c.gI = reinterpret_cast<float*>(aptr(L.gI));
    dep_run<SR, BWD, In>(c, arc + arc_off, N, (BWD && glogZ) ? glogZ[b] : 1.f, logZ + b,
                         BWD ? garc + arc_off : nullptr, tid, kThreads, [] { __syncthreads(); });
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
                      float* logZ, float* gdec, float* gatt, void* ws, size_t ws_stride, size_t lds,
                      hipStream_t stream) {
    auto k = dmv1o_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), lds, stream, (const typename In::T*)dec,
                       (const typename In::T*)attach, lengths, N, glogZ, logZ, gdec, gatt, (char*)ws, ws_stride);
    return check_launch("dmv1o_kernel");
}

template <int SR, bool BWD, typename In>
static int dispatch_dmv_mode(int mode, const void* dec, const void* attach, const int64_t* lengths, int B, int N,
                             const float* glogZ, float* logZ, float* gdec, float* gatt, void* ws, size_t ws_stride,
                             size_t lds, hipStream_t s) {
    switch (mode) {
        case 0: return launch_dmv<SR, 0, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, ws, ws_stride, lds, s);
        case 1: return launch_dmv<SR, 1, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, ws, ws_stride, lds, s);
        default: return launch_dmv<SR, 2, BWD, In>(dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, ws, ws_stride, lds, s);
    }
}

template <bool BWD>
static int run_dmv(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                   int semiring, const float* glogZ, float* logZ, float* gdec, float* gatt, void* ws, size_t ws_bytes,
                   void* stream) {
    if (B < 0 || N < 2) return set_error(VLG_ERR_SHAPE, "dmv1o: need B >= 0 and N >= 2 (got B=%d N=%d)", B, N);
    if (N > 255) return set_error(VLG_ERR_SHAPE, "dmv1o: N=%d exceeds the supported maximum of 255", N);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "dmv1o: in_dtype %d", in_dtype);
    if (semiring != VLG_SR_LOG && semiring != VLG_SR_MAX) return set_error(VLG_ERR_ARG, "dmv1o: semiring %d", semiring);
    if (!dec || !attach || !lengths || !logZ || (BWD && (!gdec || !gatt)))
        if (B > 0) return set_error(VLG_ERR_ARG, "dmv1o: null buffer");
    if (B == 0) return 0;
    const DmvLayout L(N, BWD, semiring == VLG_SR_MAX);
    const int mode = pick_mode(L);
    const size_t ws_stride = ws_bytes_per_sentence(L, mode), lds = lds_bytes(L, mode);
    if (ws_stride * (size_t)B > ws_bytes || (ws_stride && !ws))
        return set_error(VLG_ERR_WORKSPACE, "dmv1o: N=%d needs a %zu-byte workspace (got %zu); see vlg_workspace_bytes",
                         N, ws_stride * (size_t)B, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
#define VLG_GO(SRV, INV) \
    return dispatch_dmv_mode<SRV, BWD, INV>(mode, dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, ws, ws_stride, lds, s)
    if (semiring == VLG_SR_LOG) {
        if (in_dtype == VLG_F32) VLG_GO(VLG_SR_LOG, F32In);
        VLG_GO(VLG_SR_LOG, BF16In);
    }
    if (in_dtype == VLG_F32) VLG_GO(VLG_SR_MAX, F32In);
    VLG_GO(VLG_SR_MAX, BF16In);
#undef VLG_GO
}

template <int SR, int MODE, bool BWD, typename In>
static int launch_dep(const void* arc, const int64_t* lengths, int B, int N, const float* glogZ, float* logZ,
                      float* garc, void* ws, size_t ws_stride, size_t lds, hipStream_t stream) {
    auto k = deptree_kernel<SR, MODE, BWD, In>;
    if (int rc = prep(k, lds)) return rc;
    hipLaunchKernelGGL(k, dim3(B), dim3(kThreads), lds, stream, (const typename In::T*)arc, lengths, N, glogZ, logZ,
                       garc, (char*)ws, ws_stride);
    return check_launch("deptree_kernel");
}

template <int SR, bool BWD, typename In>
static int dispatch_dep_mode(int mode, const void* arc, const int64_t* lengths, int B, int N, const float* glogZ,
                             float* logZ, float* garc, void* ws, size_t ws_stride, size_t lds, hipStream_t s) {
    switch (mode) {
        case 0: return launch_dep<SR, 0, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, ws, ws_stride, lds, s);
        case 1: return launch_dep<SR, 1, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, ws, ws_stride, lds, s);
        default: return launch_dep<SR, 2, BWD, In>(arc, lengths, B, N, glogZ, logZ, garc, ws, ws_stride, lds, s);
    }
}

template <bool BWD>
static int run_dep(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring,
                   const float* glogZ, float* logZ, float* garc, void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || N < 2) return set_error(VLG_ERR_SHAPE, "deptree: need B >= 0 and N >= 2 (got B=%d N=%d)", B, N);
    if (N > 255) return set_error(VLG_ERR_SHAPE, "deptree: N=%d exceeds the supported maximum of 255", N);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "deptree: in_dtype %d", in_dtype);
    if (semiring != VLG_SR_LOG && semiring != VLG_SR_MAX) return set_error(VLG_ERR_ARG, "deptree: semiring %d", semiring);
    if (!arc || !logZ || (BWD && !garc))
        if (B > 0) return set_error(VLG_ERR_ARG, "deptree: null buffer");
    if (B == 0) return 0;
    const DepLayout L(N, BWD, semiring == VLG_SR_MAX);
    const int mode = pick_mode(L);
    const size_t ws_stride = ws_bytes_per_sentence(L, mode), lds = lds_bytes(L, mode);
    if (ws_stride * (size_t)B > ws_bytes || (ws_stride && !ws))
        return set_error(VLG_ERR_WORKSPACE, "deptree: N=%d needs a %zu-byte workspace (got %zu)", N,
                         ws_stride * (size_t)B, ws_bytes);
    hipStream_t s = (hipStream_t)stream;
#define VLG_GO(SRV, INV) \
    return dispatch_dep_mode<SRV, BWD, INV>(mode, arc, lengths, B, N, glogZ, logZ, garc, ws, ws_stride, lds, s)
    if (semiring == VLG_SR_LOG) {
        if (in_dtype == VLG_F32) VLG_GO(VLG_SR_LOG, F32In);
        VLG_GO(VLG_SR_LOG, BF16In);
    }
    if (in_dtype == VLG_F32) VLG_GO(VLG_SR_MAX, F32In);
    VLG_GO(VLG_SR_MAX, BF16In);
#undef VLG_GO
}

}  // namespace vlg

// =====================================================================================================
// C ABI (include/vlgae_amd.h)
// =====================================================================================================
extern "C" {

int vlg_dmv1o_inside(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                     int semiring, float* logZ, void* ws, size_t ws_bytes, void* stream) {
    return vlg::run_dmv<false>(dec, attach, lengths, B, N, in_dtype, semiring, nullptr, logZ, nullptr, nullptr, ws,
                               ws_bytes, stream);
}

int vlg_dmv1o_inside_outside(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                             int semiring, const float* grad_logZ, float* logZ, float* grad_dec, float* grad_attach,
                             void* ws, size_t ws_bytes, void* stream) {
    return vlg::run_dmv<true>(dec, attach, lengths, B, N, in_dtype, semiring, grad_logZ, logZ, grad_dec, grad_attach,
                              ws, ws_bytes, stream);
}

int vlg_deptree_inside(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring, float* logZ,
                       void* ws, size_t ws_bytes, void* stream) {
    return vlg::run_dep<false>(arc, lengths, B, N, in_dtype, semiring, nullptr, logZ, nullptr, ws, ws_bytes, stream);
}

int vlg_deptree_inside_outside(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring,
                               const float* grad_logZ, float* logZ, float* grad_arc, void* ws, size_t ws_bytes,
                               void* stream) {
    return vlg::run_dep<true>(arc, lengths, B, N, in_dtype, semiring, grad_logZ, logZ, grad_arc, ws, ws_bytes, stream);
}

size_t vlg_workspace_bytes(int op, int B, int N, int semiring) {
    if (B <= 0 || N < 2) return 0;
    const bool is_max = semiring == VLG_SR_MAX;
    switch (op) {
        case VLG_OP_DMV1O_INSIDE: { vlg::DmvLayout L(N, false, is_max); return vlg::ws_bytes_per_sentence(L, vlg::pick_mode(L)) * B; }
        case VLG_OP_DMV1O_INSIDE_OUTSIDE: { vlg::DmvLayout L(N, true, is_max); return vlg::ws_bytes_per_sentence(L, vlg::pick_mode(L)) * B; }
        case VLG_OP_DEPTREE_INSIDE: { vlg::DepLayout L(N, false, is_max); return vlg::ws_bytes_per_sentence(L, vlg::pick_mode(L)) * B; }
        case VLG_OP_DEPTREE_INSIDE_OUTSIDE: { vlg::DepLayout L(N, true, is_max); return vlg::ws_bytes_per_sentence(L, vlg::pick_mode(L)) * B; }
        default: return 0;
    }
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

}  // extern "C"
