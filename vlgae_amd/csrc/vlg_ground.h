// vlg_ground.h -- internal interface between vlg_align.hip (alignment maxima + arg-max) and vlg_ground.hip (the loss on them).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace vlg {

int ground_dense_split_txt(int B, int V);

// Scratch carving for vlg_grounding_loss (offsets in floats from the workspace base).
struct GroundPlan {
    size_t off_maxV, off_maxQ, off_part, off_coef, off_argV, off_argQ, off_featT, off_partial, off_parts, bytes;   // (offsets in floats)
    GroundPlan(int B, int Q, int V);
};

// Everything after the alignment kernel: the two cross-entropies, the scalar sums, the feature gradients.
int launch_grounding_tail(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marg,
                          int B, int Q, int V, int d, int in_dtype, float num_token, float w_v2t, float* ws,
                          const GroundPlan& p, float* out_sums, float* g_txt, float* g_vis, hipStream_t s);

// Scratch carving for vlg_align_reduced / _backward (gather_logit_reduced, joint.py:421-432): the forward leaves the max over
// regions, its position and the per-caption marginal sums behind for the backward.
struct ReducedPlan {
    size_t off_maxV, off_gV, off_sum, off_coef, off_argV, bytes;
    ReducedPlan(int B, int Q);
};

// logit[b,a] = sum_q marginal[b,q] * maxV[b,a,q] / sum_q marginal[b,q]; also stores the denominators.
int launch_reduced_logit(const float* maxV, const float* marg, int B, int Q, float* sums, float* logit, hipStream_t s);

// d logit -> both feature tensors through the arg-max positions (gated by the masks), with the grounding loss's kernels.
int launch_reduced_backward(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marg,
                            const float* g_logit, int B, int Q, int V, int d, int in_dtype, float* ws, const ReducedPlan& p,
                            float* g_txt, float* g_vis, hipStream_t s);

}  // namespace vlg
