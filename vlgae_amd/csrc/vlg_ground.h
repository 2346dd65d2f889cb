// vlg_ground.h -- internal interface between vlg_align.hip (alignment maxima + arg-max) and vlg_ground.hip (the loss on them).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vlg {

// Scratch carving for vlg_grounding_loss (offsets in floats from the workspace base).
struct GroundPlan {
    size_t off_maxV, off_maxQ, off_part, off_coef, off_argV, off_argQ, bytes;
    GroundPlan(int B, int Q, int V);
};

// Everything after the alignment kernel: the two cross-entropies, the scalar sums, the feature gradients.
int launch_grounding_tail(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marg,
                          int B, int Q, int V, int d, int in_dtype, float num_token, float w_v2t, float* ws,
                          const GroundPlan& p, float* out_sums, float* g_txt, float* g_vis, hipStream_t s);

}  // namespace vlg
