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
#include "vlg_dp_kernels.h"

#ifdef VLG_DP_HEADLINE_ONLY   // tools/build_variant.sh: a one-file library with only the Log / bf16 DMV1o kernels (compiles in seconds)
#define VLG_INST_FAMILY 0
#define VLG_INST_SR 0
#define VLG_INST_IN 1
#include "vlg_dp_inst.hip"
namespace vlg {
#define VLG_STUB(F, S, I, ARGS) \
    int VLG_DP_INST_NAME(F, S, I)(bool, int, const ARGS&) { return set_error(VLG_ERR_ARG, "headline-only build"); }
VLG_STUB(0, 0, 0, DmvArgs) VLG_STUB(0, 1, 0, DmvArgs) VLG_STUB(0, 1, 1, DmvArgs)
VLG_STUB(1, 0, 0, RulesArgs) VLG_STUB(1, 0, 1, RulesArgs) VLG_STUB(1, 1, 0, RulesArgs) VLG_STUB(1, 1, 1, RulesArgs)
VLG_STUB(2, 0, 0, DepArgs) VLG_STUB(2, 0, 1, DepArgs) VLG_STUB(2, 1, 0, DepArgs) VLG_STUB(2, 1, 1, DepArgs)
#undef VLG_STUB
}  // namespace vlg
#endif

namespace vlg {

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


// dispatch to the translation unit that holds the (family, semiring, input type) instantiations
#define VLG_DP_PICK(F, sr, in) \
    ((sr) == VLG_SR_LOG ? ((in) == VLG_F32 ? VLG_DP_INST_NAME(F, 0, 0) : VLG_DP_INST_NAME(F, 0, 1)) \
                        : ((in) == VLG_F32 ? VLG_DP_INST_NAME(F, 1, 0) : VLG_DP_INST_NAME(F, 1, 1)))

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
    const bool walk = BWD && semiring == VLG_SR_MAX;   // the Max semiring's outside pass is the back-pointer walk: lean layout (the kernel derives the same flag)
    const int mode = pick_mode<DmvLayout>(N, BWD, semiring == VLG_SR_MAX, kLdsBudget, walk);
    const DmvLayout L(N, BWD, semiring == VLG_SR_MAX, mode, walk);
    const size_t ws_stride = L.ws_bytes, lds = L.lds_bytes;
    if (ws_stride * (size_t)B > ws_bytes || (ws_stride && !ws))
        return set_error(VLG_ERR_WORKSPACE, "dmv1o: N=%d needs a %zu-byte workspace (got %zu); see vlg_workspace_bytes",
                         N, ws_stride * (size_t)B, ws_bytes);
    const DmvArgs a{dec, attach, lengths, B, N, glogZ, logZ, gdec, gatt, heads, ws, ws_stride, lds, (hipStream_t)stream};
    return VLG_DP_PICK(0, semiring, in_dtype)(BWD, mode == 0 && N <= kShortN ? kModeShort : mode, a);   // short sentences: the smaller code image
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
    const DepArgs a{arc, lengths, B, N, glogZ, logZ, garc, heads, ws, ws_stride, lds, (hipStream_t)stream};
    return VLG_DP_PICK(2, semiring, in_dtype)(BWD, mode == 0 && N <= kShortN ? kModeShort : mode, a);
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

int vlg_dmv1o_viterbi(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                      const float* grad_best, float* best_score, float* grad_dec, float* grad_attach, int64_t* heads, void* ws,
                      size_t ws_bytes, void* stream) {
    if (B > 0 && !heads && !grad_attach) return vlg::set_error(VLG_ERR_ARG, "dmv1o_viterbi: pass grad_attach and / or heads");
    if (B > 0 && grad_dec && !grad_attach) return vlg::set_error(VLG_ERR_ARG, "dmv1o_viterbi: grad_dec needs grad_attach");
    return vlg::run_dmv<true>(dec, attach, lengths, B, N, in_dtype, VLG_SR_MAX, grad_best, best_score, grad_dec, grad_attach,
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
    const bool walk = bwd && is_max;
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
    return VLG_DP_PICK(1, semiring, in_dtype)(bwd, mode == 0 && N <= kShortN ? kModeShort : mode, a);
}

size_t vlg_workspace_bytes(int op, int B, int N, int semiring) {
    if (B <= 0 || N < 2) return 0;
    const bool is_max = semiring == VLG_SR_MAX;
    using namespace vlg;
    switch (op) {
        case VLG_OP_DMV1O_INSIDE:
            return DmvLayout(N, false, is_max, pick_mode<DmvLayout>(N, false, is_max, kLdsBudget)).ws_bytes * B;
        case VLG_OP_DMV1O_INSIDE_OUTSIDE:   // Log: the replay layout; Max: the walk layout (the one-hot replay is not compiled any more, round 4)
            if (is_max) return DmvLayout(N, true, true, pick_mode<DmvLayout>(N, true, true, kLdsBudget, true), true).ws_bytes * B;
            return DmvLayout(N, true, false, pick_mode<DmvLayout>(N, true, false, kLdsBudget)).ws_bytes * B;
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
