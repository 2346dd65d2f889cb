// vlg_capi.cpp -- thread-local error state and version of the C ABI (include/vlgae_amd.h).
#include "vlg_common.h"

namespace vlg {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace vlg

extern "C" {
const char* vlg_last_error(void) { return vlg::error_buffer(); }
int vlg_version(void) { return 144; }   // round 6: vlg_ff_linear_act_backward takes k (256 / 512 / 32) and w_kn, + vlg_ff_linear_mlp_act_backward, vlg_ff_linear_kn, vlg_ff_transpose256 over a pointer array; 143: + vlg_ff_linear_act / _backward (fused row-streaming Linear + element-wise pass); 142: vlg_attn_fuse key_chunk + workspace + grad_dtype
}
