// vlg_dp_pair.hip -- marginals AND the Viterbi tree of one batch of potentials in a single launch (dmv1o_pair_kernel,
// vlg_dp_kernels.h): what `DependencyBoxRel.lang_feat_max_tree` asks of DMV1o every step (src/model/joint.py:251-258: the partition's
// gradient for the arc marginals, the argmax for the predicted heads) and, when the parser trains on the Viterbi tree
// (`-DMV1o(...).max.sum()`, src/model/ldndmv.py:277-281), the tree counts of the same potentials as well.
#include "vlg_dp_kernels.h"

namespace vlg {

namespace {

// both placements must be the everything-in-LDS one (mode 0) and fit one CU TOGETHER: the two workgroups of a sentence run side by side
bool pair_ok(int N, size_t* lds) {
    const DmvLayout Llog(N, true, false, 0, false), Lmax(N, true, true, 0, true);
    *lds = Llog.lds_bytes > Lmax.lds_bytes ? Llog.lds_bytes : Lmax.lds_bytes;
    return pick_mode<DmvLayout>(N, true, false, kLdsBudget, false) == 0 && pick_mode<DmvLayout>(N, true, true, kLdsBudget, true) == 0 &&
           2 * *lds <= kLdsBudget;
}

template <typename In>
int launch_pair(const void* dec, const void* attach, const int64_t* lengths, int B, int N, size_t lds, float* logZ, float* gdec_log,
                float* gatt_log, float* best, float* gdec_max, float* gatt_max, int64_t* heads, hipStream_t s) {
    if (N <= kShortN) {   // the short-sentence code image (vlg_dp_core.h: kSpansShort)
        auto k = dmv1o_pair_kernel<In, kModeShort>;
        if (int rc = prep(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(B, 2), dim3(kThreads), lds, s, (const typename In::T*)dec, (const typename In::T*)attach, lengths, N, logZ,
                           gdec_log, gatt_log, best, gdec_max, gatt_max, (long long*)heads);
    } else {
        auto k = dmv1o_pair_kernel<In, 0>;
        if (int rc = prep(k, lds)) return rc;
        hipLaunchKernelGGL(k, dim3(B, 2), dim3(kThreads), lds, s, (const typename In::T*)dec, (const typename In::T*)attach, lengths, N, logZ,
                           gdec_log, gatt_log, best, gdec_max, gatt_max, (long long*)heads);
    }
    return check_launch("dmv1o_pair_kernel");
}

}  // namespace

}  // namespace vlg

extern "C" {

int vlg_dmv1o_marginals_viterbi_supported(int N) {
    size_t lds;
    return N >= 2 && N <= 255 && vlg::pair_ok(N, &lds) ? 1 : 0;
}

int vlg_dmv1o_marginals_viterbi(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype, float* logZ,
                                float* grad_dec, float* grad_attach, float* best_score, float* tree_dec, float* tree_attach, int64_t* heads,
                                void* stream) {
    using namespace vlg;
    if (B < 0 || N < 2) return set_error(VLG_ERR_SHAPE, "dmv1o_marginals_viterbi: need B >= 0 and N >= 2 (got B=%d N=%d)", B, N);
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "dmv1o_marginals_viterbi: in_dtype %d", in_dtype);
    size_t lds = 0;
    if (N > 255 || !pair_ok(N, &lds))
        return set_error(VLG_ERR_SHAPE, "dmv1o_marginals_viterbi: N=%d -- the two passes do not share a CU's LDS (see vlg_dmv1o_marginals_viterbi_supported); "
                         "launch vlg_dmv1o_inside_outside and vlg_dmv1o_viterbi on two streams", N);
    if (B == 0) return 0;
    if (!dec || !attach || !lengths || !logZ || !grad_attach || !best_score || !heads)
        return set_error(VLG_ERR_ARG, "dmv1o_marginals_viterbi: null buffer");
    if (tree_dec && !tree_attach) return set_error(VLG_ERR_ARG, "dmv1o_marginals_viterbi: tree_dec needs tree_attach");
    hipStream_t s = (hipStream_t)stream;
    return in_dtype == VLG_F32 ? launch_pair<F32In>(dec, attach, lengths, B, N, lds, logZ, grad_dec, grad_attach, best_score, tree_dec, tree_attach, heads, s)
                               : launch_pair<BF16In>(dec, attach, lengths, B, N, lds, logZ, grad_dec, grad_attach, best_score, tree_dec, tree_attach, heads, s);
}

}  // extern "C"
