// vlg_dp_inst.hip -- the kernel instantiations of one (family, semiring, input type) of the structured DP.
// Compiled 12 times by vlgae_amd/build.py with -DVLG_INST_FAMILY={0,1,2} -DVLG_INST_SR={0,1} -DVLG_INST_IN={0,1}
// (see vlg_dp_kernels.h); each object holds the 2 (inside / fused) x 5 (four placement modes + the short-sentence code image of placement 0) kernels of its combination.
#include "vlg_dp_kernels.h"

#if !defined(VLG_INST_FAMILY) || !defined(VLG_INST_SR) || !defined(VLG_INST_IN)
#error "vlg_dp_inst.hip: define VLG_INST_FAMILY, VLG_INST_SR and VLG_INST_IN"
#endif

namespace vlg {

#if VLG_INST_IN == 0
using InstIn = F32In;
#else
using InstIn = BF16In;
#endif

#if VLG_INST_FAMILY == 0
using InstArgs = DmvArgs;
#define VLG_INST_LAUNCH launch_dmv
#elif VLG_INST_FAMILY == 1
using InstArgs = RulesArgs;
#define VLG_INST_LAUNCH launch_rules
#else
using InstArgs = DepArgs;
#define VLG_INST_LAUNCH launch_dep
#endif

template <bool BWD>
static int by_mode(int mode, const InstArgs& a) {
#ifdef VLG_DP_HEADLINE_ONLY   // tools/ A/B builds: the all-in-LDS and overlay placements only
    if (mode > 1 && mode != kModeShort) return set_error(VLG_ERR_ARG, "headline-only build");
    if (mode == kModeShort) return VLG_INST_LAUNCH<VLG_INST_SR, kModeShort, BWD, InstIn>(a);
    return mode == 0 ? VLG_INST_LAUNCH<VLG_INST_SR, 0, BWD, InstIn>(a) : VLG_INST_LAUNCH<VLG_INST_SR, 1, BWD, InstIn>(a);
#else
    switch (mode) {
        case 0: return VLG_INST_LAUNCH<VLG_INST_SR, 0, BWD, InstIn>(a);
        case 1: return VLG_INST_LAUNCH<VLG_INST_SR, 1, BWD, InstIn>(a);
        case 2: return VLG_INST_LAUNCH<VLG_INST_SR, 2, BWD, InstIn>(a);
        case kModeShort: return VLG_INST_LAUNCH<VLG_INST_SR, kModeShort, BWD, InstIn>(a);
        default: return VLG_INST_LAUNCH<VLG_INST_SR, 3, BWD, InstIn>(a);
    }
#endif
}

int VLG_DP_INST_NAME(VLG_INST_FAMILY, VLG_INST_SR, VLG_INST_IN)(bool bwd, int mode, const InstArgs& a) {
    return bwd ? by_mode<true>(mode, a) : by_mode<false>(mode, a);
}

}  // namespace vlg
