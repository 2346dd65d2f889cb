/*
 * vlgae_amd.h -- C ABI of the MI355X-native structured-DP hot path of VLGAE.
 *
 * The reference (LouChao98/VLGAE) is pure Python/PyTorch and has NO foreign-function interface
 * for this path; the entry points below are what a binding for it would bind.  Each one cites
 * the reference interface it replaces (paths relative to the reference checkout).  The host
 * side that mirrors the reference's Python API on top of this ABI is vlgae_amd/torch_struct/
 * (ctypes; see INTEGRATION.md for the exact stub a maintainer adds).
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer to a contiguous, caller-owned buffer (in practice the
 *     data_ptr() of a PyTorch-ROCm tensor).  The library allocates and frees nothing.
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  Calls only
 *     enqueue work: no host synchronisation, no hidden allocation -> HIP-graph capturable.
 *   - Thread-safe / re-entrant (PyTorch runs backward on its own thread): no global mutable
 *     state; the last error message is thread-local.
 *   - Return 0 on success; a VLG_ERR_* code (> 0x1000) or a hipError_t value otherwise, with a
 *     human-readable message available from vlg_last_error().  Nothing throws across the ABI.
 *   - Input element type: VLG_F32 or VLG_BF16 (`in_dtype`).  Charts, log-sum-exp accumulators
 *     and all outputs are fp32.
 *   - `lengths[b]` = number of words of sentence b (root excluded), 1 <= len <= N-1.  A sentence
 *     with an out-of-range length gets logZ = NaN and all-zero gradients.
 *   - Semiring zero is the finite sentinel -1e12 (semirings.py:16,128), never -inf.  Potentials MAY contain -inf (or
 *     anything below -1e30): they are clamped to -1e30 as they are loaded, i.e. treated as probability zero like the
 *     reference's logsumexp does; +inf / NaN potentials are the caller's error (NaN is clamped to the floor as well).
 *   - Every pointer of one call must belong to the CURRENT HIP device of the calling thread (launches go there).
 */
#ifndef VLGAE_AMD_H
#define VLGAE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { VLG_F32 = 0, VLG_BF16 = 1 };
enum { VLG_SEMIRING_LOG = 0, VLG_SEMIRING_MAX = 1 };
enum {
    VLG_OP_DMV1O_INSIDE = 0,
    VLG_OP_DMV1O_INSIDE_OUTSIDE = 1,
    VLG_OP_DEPTREE_INSIDE = 2,
    VLG_OP_DEPTREE_INSIDE_OUTSIDE = 3
};
enum {
    VLG_ERR_SHAPE = 0x1001,     /* mirrors the reference's shape asserts (deptree.py:149,154) */
    VLG_ERR_DTYPE = 0x1002,
    VLG_ERR_ARG = 0x1003,
    VLG_ERR_WORKSPACE = 0x1004
};

/* DMV1oStruct._dp through _Struct.sum -- src/model/torch_struct/dmv.py:19-66, helpers.py:101-116
 * (what `DMV1o([dec, attach], lengths).partition` / `.max` evaluate, distributions.py:116-124,190-193).
 *   dec    [B,N,2(dir),2(valence),2(decision)]   attach [B,N(head),N(child),2(valence)]   (root-merged)
 *   logZ   [B]  log-partition (semiring 0) or Viterbi score (semiring 1)
 *   ws     workspace of vlg_workspace_bytes(VLG_OP_DMV1O_INSIDE, B, N, semiring) bytes (0 for N <= ~100). */
int vlg_dmv1o_inside(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                     int semiring, float* logZ, void* ws, size_t ws_bytes, void* stream);

/* Inside + outside in ONE launch.  Replaces `torch.autograd.grad(dist.partition.sum(), [dec, attach])`
 * (src/model/joint.py:255,320; ldndmv.py:269,295; dmv.py:121), `_Struct.marginals` (helpers.py:118-157)
 * and the backward of `-dist.partition.sum()` / `-dist.max.sum()` (ldndmv.py:277-281).
 *   grad_logZ   [B] upstream gradient of logZ, or NULL for all-ones
 *   grad_dec    [B,N,2,2,2], grad_attach [B,N,N,2]: d(sum_b grad_logZ[b]*logZ[b]) / d(potentials) =
 *               posterior expected counts (Log) or the 0/1 indicator of the best tree (Max; first
 *               arg-max on ties like torch.max).  Fully written, exact zeros at padded positions.
 *               grad_dec may be NULL (attach counts only -- what `.marginals` / `.argmax` keep, dmv.py:68-69); with the
 *               Max semiring the tree is then read off the back-pointers instead of replaying the outside pass. */
int vlg_dmv1o_inside_outside(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                             int semiring, const float* grad_logZ, float* logZ, float* grad_dec, float* grad_attach,
                             void* ws, size_t ws_bytes, void* stream);

/* DepTree._dp -- src/model/torch_struct/deptree.py:25-76 (`DependencyCRF(arc, lengths).partition/.max`).
 *   arc [B,N,N] head -> child scores, root at index 0; lengths may be NULL (= N-1, deptree.py:151-152). */
int vlg_deptree_inside(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring, float* logZ,
                       void* ws, size_t ws_bytes, void* stream);

/* `DependencyCRF(...).marginals` / `.argmax` -- distributions.py:126-133,162-174 via helpers.py:118-157.
 *   grad_arc [B,N,N] arc marginals (Log) or the 0/1 best tree (Max). */
int vlg_deptree_inside_outside(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, int semiring,
                               const float* grad_logZ, float* logZ, float* grad_arc, void* ws, size_t ws_bytes,
                               void* stream);

/* Best tree as a head vector, on the device -- replaces the callers' `dist.argmax.sum(-1).nonzero()` + scatter
 * (src/model/ldndmv.py:301-303, joint.py:256-258, dmv.py:127-129), which synchronises the host every step.
 *   heads [B,N] int64: heads[b,c] = head of word c in the Viterbi tree (0 = the root token), 0 for c = 0 and padding
 *   best_score [B]: the Max-semiring value (`dist.max`).  Workspace as for VLG_OP_*_INSIDE_OUTSIDE, semiring 1. */
int vlg_dmv1o_decode(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                     float* best_score, int64_t* heads, void* ws, size_t ws_bytes, void* stream);

/* Same for the projective CRF; with `arc` = arc marginals this is the MBR decode of ldndmv.py:294-299. */
int vlg_deptree_decode(const void* arc, const int64_t* lengths, int B, int N, int in_dtype, float* best_score,
                       int64_t* heads, void* ws, size_t ws_bytes, void* stream);

/* The DMV DP fed directly from the scorer's rule tables.  Folds into the kernel's load
 * stage what DiscriminativeNDMV._forward does between the scorer and the DP (src/model/ldndmv.py:189-209):
 *   attach[b,h,c,v] = attach_rule[b, h, token[b,c], dir(h,c), v]   (gather by the child's token + tril/triu select)
 *                   = mask_fill where head_mask[b,h]                (function_mask, ldndmv.py:194-198)
 *   root[b,c]       = root_rule[(b,) token[b,c]]                    (ldndmv.py:207)
 *   DMV1o.merge(dec, attach, root)                                  (ldndmv.py:209)
 * and returns the expected counts in RULE space (the adjoint of those gathers):
 *   attach_rule [B,L,T,2(dir),2(val)], dec [B,L,2,2,2], root_rule [T] (root_per_sentence = 0) or [B,T] (= 1),
 *   token [B,L] int64 in [0,T), head_mask [B,L] uint8 or NULL, lengths [B];
 *   logZ [B]; grad_rule [B,L,T,2,2], grad_dec [B,L,2,2,2], grad_root [B,T] (all three or none; zero-filled here);
 *   heads [B,L+1] optional (Viterbi heads: semiring must be 1, else VLG_ERR_ARG).  A sentence with a token id outside
 *   [0,T) is treated like one with an out-of-range length: logZ = NaN, zero counts.  Workspace: vlg_workspace_bytes(VLG_OP_DMV1O_*, B, L+1, semiring). */
int vlg_dmv1o_rules(const void* attach_rule, const void* dec, const void* root_rule, int root_per_sentence,
                    const int64_t* token, const uint8_t* head_mask, const int64_t* lengths, int B, int L, int T,
                    int in_dtype, int semiring, float mask_fill, const float* grad_logZ, float* logZ, float* grad_rule,
                    float* grad_dec, float* grad_root, int64_t* heads, void* ws, size_t ws_bytes, void* stream);

/* DMV1o.merge -- src/model/torch_struct/distributions.py:253-265.
 *   dec [B,L,2,2,2], attach [B,L,L,2], root [B,L]  ->  dec_wroot [B,L+1,2,2,2], attach_wroot [B,L+1,L+1,2]
 *   (always fp32, like the reference's torch.full). */
int vlg_dmv1o_merge(const void* dec, const void* attach, const void* root, int B, int L, int in_dtype, float one,
                    float zero, float* dec_wroot, float* attach_wroot, void* stream);

/* Batch sum of the expected counts, out[N*8 + N*N*2] = sum_b [grad_dec[b] | grad_attach[b]]: the gradient of
 * position-tied potentials, i.e. the marginal-loss gradient that the data-parallel all-reduce carries
 * (Lightning DDP in the reference, config/trainer/train.yaml:27-29; bench.py's multi-GPU step).  Fixed summation order. */
int vlg_dmv1o_count_sum(const float* grad_dec, const float* grad_attach, int B, int N, float* out, void* stream);

/* Bytes of caller-provided scratch the op needs for (B, N); 0 when the charts fit in LDS. */
size_t vlg_workspace_bytes(int op, int B, int N, int semiring);

/* Region x word bilinear alignment -- DependencyBoxRel.gather_logit_simple, src/model/joint.py:406-419:
 *   attmap[b,a,q,v] = sum_k txt[b,q,k] * vis[a,v,k];  = neg_inf where !vmask[a,v] or !tmask[b,q]
 *   txt [B,Q,d], vis [A,V,d] (in_dtype), tmask [B,Q] / vmask [A,V] uint8 (NULL = all true).
 * Any subset of the outputs may be requested (NULL = skip):
 *   out_full [B,A,Q,V] fp32           (the reference's return value)
 *   out_maxV [B,A,Q]   max over V     (consumer: loss_grounding_factor_ce, joint.py:473-476)
 *   out_maxQ [B,A,V]   max over Q     (joint.py:478-483)
 *   out_diag [B,Q,V]   attmap[b,b]    (decode_grounding_on_factor, joint.py:519-524; needs A == B) */
int vlg_bilinear_align(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, int B, int A,
                       int Q, int V, int d, int in_dtype, float neg_inf, float* out_full, float* out_maxV,
                       float* out_maxQ, float* out_diag, void* stream);

/* Backward of the materialised alignment tensor -- what autograd runs for joint.py:413-418 when the reference's own
 * loss_grounding_factor_ce consumes gather_logit_simple's [B,A,Q,V] output (loss.backward()):
 *   grad_txt[b,q,:] = tmask[b,q] * sum_{a,v} grad_out[b,a,q,v] * vmask[a,v] * vis[a,v,:]
 *   grad_vis[a,v,:] = vmask[a,v] * sum_{b,q} grad_out[b,a,q,v] * tmask[b,q] * txt[b,q,:]      (masked_fill_ passes no gradient)
 *   grad_out [B,A,Q,V] fp32; txt [B,Q,d], vis [A,V,d] (in_dtype); grad_txt [B,Q,d], grad_vis [A,V,d] fp32 (either may be NULL).
 * d in {32, 64, 128}.  The cotangent is read in place with both masks fused.  With d = 128 and at most 96 rows / contraction
 * positions per pair (config-2) the products run on the bf16 matrix cores: the fp32 cotangent is split on the fly into two bf16
 * terms (g = t0 + t1, dropping < 2^-17 |g|), bf16 features are used as they are, fp32 features as two bf16 parts (x = hi + lo;
 * the lo x t1 product, < 2^-16 of the term, is dropped), against contraction-major copies of the features that live in `ws`:
 * vlg_bilinear_align_backward_workspace(...) bytes of device scratch (0 when the fast path does not apply; ws may then be NULL).
 * Other shapes: fp32 matrix-core products, exact (environment VLG_BWD_F32_EXACT=1 forces that path for fp32 features). */
size_t vlg_bilinear_align_backward_workspace(int B, int A, int Q, int V, int d, int in_dtype);
int vlg_bilinear_align_backward(const float* grad_out, const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask,
                                int B, int A, int Q, int V, int d, int in_dtype, void* ws, size_t ws_bytes, float* grad_txt,
                                float* grad_vis, void* stream);

/* The grounding loss on the alignment, without the [B,A,Q,V] tensor -- DependencyBoxRel.gather_logit_simple followed by
 * loss_grounding_factor_ce, src/model/joint.py:406-419 + 439-491, A == B:
 *   att[b,a,q,v] = <txt[b,q], vis[a,v]>, masked -> neg_inf; on the pairs a == b the POS prior subtracts
 *   pen[b,q,seg_of_v[v]] (joint.py:446-470: pen = 100 for every named factor whose POS set holds the token's tag and
 *   whose segment is not seg(v); NULL = no prior);
 *   txt2vis = - sum marginal[b,q] * log_softmax_a(max_v att)[b,b,q]     (joint.py:472-476)
 *   vis2txt = - sum vmask[a,v]    * log_softmax_b(max_q att)[a,a,v]     (joint.py:478-483)
 *   total   = txt2vis / (txt2vis + 1e-6) * num_token + w_vis2txt * vis2txt / (vis2txt + 1e-6) * num_token
 *             (denominators detached, joint.py:477,484-489; w_vis2txt <= 0 drops the second term like `if vis2txt > 0`)
 *   out_sums[3] = {txt2vis, vis2txt, total} (device, fp32); g_txt [B,Q,d], g_vis [B,V,d] = d total / d features (fp32,
 *   both or either may be NULL).  The gradient passes through the first arg-max of each maximum and only where both
 *   masks are on.  txt, vis in in_dtype; marginal [B,Q] fp32; masks uint8 or NULL; d in {32, 64, 128}.
 *   ws: vlg_grounding_loss_workspace(B, Q, V) bytes.  No atomics: results are bit-reproducible.
 *   Round 5, in_dtype = VLG_F32 with d = 128 (the reference's `precision: 32`, config/trainer/train.yaml:20): the alignment's maxima and
 *   positions are taken on two fp16 parts per feature under one power-of-two scale per tensor, three matrix-core products per pair
 *   (2^-22 relative: float32's own level; 1.18 -> 0.36 ms at B = 256, Q = 82, V = 36); the workspace holds the split features. */
size_t vlg_grounding_loss_workspace(int B, int Q, int V);
int vlg_grounding_loss(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marginal,
                       const float* pen, const uint8_t* seg_of_v, int n_seg, int B, int Q, int V, int d, int in_dtype,
                       float neg_inf, float num_token, float w_vis2txt, void* ws, size_t ws_bytes, float* out_sums,
                       float* g_txt, float* g_vis, void* stream);

/* gather_logit_reduced, src/model/joint.py:421-432 (the input of loss_grounding_cap_img_ll :493-499 and
 * decode_grounding_on_image :506-510), for B captions x B images without the [B,B,Q,V] tensor:
 *   out_logit[b,a] = sum_q marginal[b,q] * max_v <txt[b,q], vis[a,v]> / sum_q marginal[b,q]      (masked entries = neg_inf)
 * The forward leaves the maxima, their positions and the denominators in ws (vlg_align_reduced_workspace(B, Q) bytes);
 * vlg_align_reduced_backward turns g_logit [B,B] into g_txt [B,Q,d] / g_vis [B,V,d] (fp32, either may be NULL) through the
 * first arg-max of each maximum, only where both masks are on (masked_fill_, joint.py:417-418).  It reads the forward's
 * ws and may be called more than once.  marginal [B,Q] fp32 is a constant (no gradient).  d in {32, 64, 128}. */
size_t vlg_align_reduced_workspace(int B, int Q);
int vlg_align_reduced(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marginal, int B,
                      int Q, int V, int d, int in_dtype, float neg_inf, void* ws, size_t ws_bytes, float* out_logit, void* stream);
int vlg_align_reduced_backward(const void* txt, const void* vis, const uint8_t* tmask, const uint8_t* vmask, const float* marginal,
                               const float* g_logit, int B, int Q, int V, int d, int in_dtype, void* ws, size_t ws_bytes,
                               float* g_txt, float* g_vis, void* stream);

/* The tensor half of decode_grounding_on_factor, src/model/joint.py:512-596, on the fused
 * outputs of vlg_bilinear_align (out_diag -> logit, out_maxV -> maxV):
 *   logit [B,Q,V] fp32 is edited IN PLACE like the reference's diagonal copy: minus pen[b,q,seg_of_v[v]] (POS prior,
 *   :528-552; pen / seg_of_v NULL = off), then with use_heuristic (:554-594): a query row whose best column is one of
 *   the n_box box columns [0, n_box) (and is > -1e5) selects that box for its sentence; relation columns
 *   rel_offset + i * n_box + j lose 100 unless both boxes are selected (rows >= n_word_rows do not vote) and are set
 *   to -1e10 on i == j; attribute columns attr_offset + i are set to -1e10 unless box i is selected (offsets -1 = no
 *   such block).  top5 [B,Q,5] int32 = the five best columns of every row in descending order (:596; equal values by
 *   ascending column, -1 past V); factor2img [B,Q] int32 = first arg-max over a of maxV [B,A,Q] (:520; both or
 *   neither NULL).  ws (vlg_grounding_decode_workspace(B, n_box) bytes) is optional: with it, batches smaller than the
 *   chip spread each sentence's rows over several workgroups (two launches); NULL = one workgroup per sentence. */
size_t vlg_grounding_decode_workspace(int B, int n_box);
int vlg_grounding_decode(float* logit, const float* pen, const uint8_t* seg_of_v, int n_seg, int B, int Q, int V,
                         int use_heuristic, int n_box, int rel_offset, int attr_offset, int n_word_rows, const float* maxV,
                         int A, int32_t* factor2img, int32_t* top5, void* ws, size_t ws_bytes, void* stream);

/* The arc encoder's trilinear term -- lang_feat word+maxdep, src/model/joint.py:281-284:
 *   out[m,h] = sum_{x,y} child[m,x] * w[x,h,y] * parent[m,y]        (m = flattened batch x position)
 *   child [M,X], parent [M,Y], w [X,H,Y] in in_dtype; out [M,H] fp32.  Y in {32, 64, 128}; H a multiple of 16, <= 128;
 *   X <= 256.  The [M,H,Y] intermediate torch.einsum materialises is never built. */
int vlg_trilinear(const void* child, const void* w, const void* parent, int M, int X, int H, int Y, int in_dtype, float* out,
                  void* stream);

/* The same with a caller-owned scratch for fixed-order partial sums (vlg_trilinear_workspace bytes; 0 = none needed): for bf16 and
 * H = Y = 128 the x range is then split over ~one workgroup per CU (slabs added in a fixed order by a second launch) instead of
 * two ranges met by atomics.  ws may be NULL / too small: the call then behaves like vlg_trilinear.
 * Round 5, in_dtype = VLG_F32 with H = Y = 128, M >= 1024: with the workspace the product runs on two fp16 parts per operand row under
 * power-of-two row scales (three v_mfma_f32_16x16x32_f16 per pair, 2^-22 relative) instead of v_mfma_f32_16x16x4_f32: 444 -> 145 us at
 * M = 10 496, closer to float64 than the fp32 MFMAs; vlg_trilinear_backward(_g) likewise for X = H = Y = 128 (its workspace query covers it). */
size_t vlg_trilinear_workspace(int M, int X, int H, int Y, int in_dtype);
int vlg_trilinear_ws(const void* child, const void* w, const void* parent, int M, int X, int H, int Y, int in_dtype, void* ws,
                     size_t ws_bytes, float* out, void* stream);

/* Adjoint of vlg_trilinear for the cotangent g [M,H] (fp32): d_child [M,X], d_w [X,H,Y], d_parent [M,Y], all fp32, each
 * optional (NULL = skip).  The two input gradients are the forward kernel run on permuted copies of w; d_w contracts over
 * m from transposed copies of the three operands.  H and Y in {32, 64, 128}; X a multiple of 16, <= 128 (for d_child).
 * With bf16 operands the cotangent (and the product child * g inside d_w) is rounded to bf16, like any bf16 autograd.
 * ws: vlg_trilinear_backward_workspace(M, X, H, Y, in_dtype) bytes. */
size_t vlg_trilinear_backward_workspace(int M, int X, int H, int Y, int in_dtype);
int vlg_trilinear_backward(const void* child, const void* w, const void* parent, const float* g, int M, int X, int H, int Y,
                           int in_dtype, void* ws, size_t ws_bytes, float* d_child, float* d_w, float* d_parent, void* stream);
/* The same with the cotangent in `g_dtype` (VLG_BF16 with bf16 features: taken as it is, no fp32 round trip). */
int vlg_trilinear_backward_g(const void* child, const void* w, const void* parent, const void* g, int g_dtype, int M, int X, int H,
                             int Y, int in_dtype, void* ws, size_t ws_bytes, float* d_child, float* d_w, float* d_parent, void* stream);

/* Attention-fuse that feeds the parser -- DependencyBoxRel._forward, src/model/joint.py:670-674:
 *   att = softmax_v(vis[b] . txt[b,1:]) ; x = att . vis_mid[b] ; out = LayerNorm(enc_x + x) * gamma + beta
 *   vis [B,V,d], txt [B,L+1,d] (root slot first, skipped), vis_mid [B,V,h], enc_x [B,L,h] (in_dtype);
 *   gamma, beta [h] fp32; out [B,L,h] fp32; out_att [B,L,V] fp32 optional (NULL = skip).
 *   key_chunk: 0 = automatic -- one pass over the keys for V <= 256 (tens of regions), the key-split form above that (the shipped
 *   factor layout of config/model/vlgae.yaml:40-42 has V = 36 + 36^2 + 36 + 1 = 1369 keys per image): chunks of keys per wavefront,
 *   streaming-softmax records merged in chunk order; > 0 = that many keys per chunk (rounded up to a multiple of 64; >= V: one pass).
 *   ws: vlg_attn_fuse_workspace(B, L, V, h, key_chunk) bytes of device scratch (0 when the keys are not split; NULL is fine then).
 *   saved (optional): vlg_attn_fuse_saved_bytes(...) bytes (0 when the keys are not split) that receive the merged streaming-softmax
 *   records of every (sentence, 16-word tile); handed to vlg_attn_fuse_backward they spare it the recomputation of the forward. */
size_t vlg_attn_fuse_workspace(int B, int L, int V, int h, int key_chunk);
size_t vlg_attn_fuse_saved_bytes(int B, int L, int V, int h, int key_chunk);
int vlg_attn_fuse(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                  const float* beta, int B, int L, int V, int d, int h, int in_dtype, float eps, int key_chunk, void* ws,
                  size_t ws_bytes, float* saved, float* out_att, float* out, void* stream);

/* Adjoint of vlg_attn_fuse -- what autograd derives for src/model/joint.py:670-674 (the fuse sits inside
 * DependencyBoxRel._forward, so training back-propagates through it into the feature encoders and the LayerNorm).
 *   dout [B,L,h] fp32 = cotangent of `out`, its rows at dout + b ld_dout_b + l ld_dout_l (elements; contiguous: L h and h; the
 *   parser's context_mode 'mean' sends every position of a sentence the same row: h and 0, nothing materialised);
 *   inputs as in vlg_attn_fuse (in_dtype), gamma [h] fp32; key_chunk as in vlg_attn_fuse; saved: what vlg_attn_fuse wrote for the
 *   same inputs and key_chunk, or NULL (the forward's records are then recomputed: same results).
 *   d_vis [B,V,d], d_txt [B,L+1,d] (root slot row = 0), d_vis_mid [B,V,h], d_enc_x [B,L,h] in grad_dtype (VLG_F32, or VLG_BF16 with
 *   bf16 inputs: rounded to nearest even once from the fp32 accumulators -- the bits an fp32 result and a cast give); d_gamma [h],
 *   d_beta [h] fp32; all written (no accumulation).  Needs d, h multiples of 16 and <= 256 (VLG_ERR_SHAPE otherwise).
 *   ws: vlg_attn_fuse_backward_workspace(B, L, V, d, h, grad_dtype, key_chunk) bytes of device scratch (softmax / score-gradient
 *   tiles, partial LayerNorm-parameter sums, chunk records).  Bit-reproducible: no atomics, fixed summation orders. */
size_t vlg_attn_fuse_backward_workspace(int B, int L, int V, int d, int h, int grad_dtype, int key_chunk);
int vlg_attn_fuse_backward(const void* vis, const void* txt, const void* vis_mid, const void* enc_x, const float* gamma,
                           const float* dout, long long ld_dout_b, long long ld_dout_l, int B, int L, int V, int d, int h, int in_dtype,
                           float eps, int key_chunk, int grad_dtype, const float* saved, void* ws, size_t ws_bytes, void* d_vis,
                           void* d_txt, void* d_vis_mid, void* d_enc_x, float* d_gamma, float* d_beta, void* stream);

/* Pairwise relation features of the visual encoder -- VisBoxRelSimpleEncoder.forward, src/model/vis_encoder/box_rel.py:41-45:
 *   rel[b,i,j,:] = LeakyReLU( rel_fc.linear( (inputs[b,i] + inputs[b,j]) / 2 ) )
 * By linearity of the Linear layer this is LeakyReLU((y[b,i] + y[b,j]) / 2 + bias) with y = inputs W^T (one library GEMM,
 * done by the caller); the [B,R,R,n_in] pairwise-mean tensor and the 35x larger GEMM over it never exist.
 *   y [B,R,H], out [B,R,R,H] (both `dtype`: VLG_F32 or VLG_BF16), bias [H] fp32 (NULL = 0), slope = LeakyReLU's (0.01).
 *   H a multiple of 4 with H/4 dividing 256. */
int vlg_box_rel_pairwise(const void* y, const float* bias, int B, int R, int H, int dtype, float slope, void* out, void* stream);

/* Its adjoint: grad_out [B,R,R,H] (`dtype`) -> grad_y [B,R,H], grad_bias [H] (fp32; grad_bias may be NULL).
 * ws: vlg_box_rel_pairwise_backward_workspace bytes (only needed with grad_bias).  Fixed summation order. */
size_t vlg_box_rel_pairwise_backward_workspace(int B, int R, int H);
int vlg_box_rel_pairwise_backward(const void* y, const float* bias, const void* grad_out, int B, int R, int H, int dtype, float slope,
                                  void* ws, size_t ws_bytes, float* grad_y, float* grad_bias, void* stream);

/* Weight / bias gradient of an nn.Linear over all token rows -- `MLP.linear` (src/model/nn/common.py:30,47-51) under
 * loss.backward(), i.e. the word / child / parent encoders of src/model/joint.py:270-277 and `vis_mlp_pre_matching`
 * (joint.py:136-138,175):   d_weight[M(out), N(in)] = dy^T x,   d_bias[M] = sum_rows dy.
 *   dy [K, ld_dy] (first M columns used), x [K, ld_x] (first N columns used): `in_dtype`, row-major, K = B*N token rows;
 *   M and N multiples of 8 (64 x 64 output tiles, partial at the edges), row strides multiples of 8 elements, dy / x / ws 16-byte aligned.
 *   d_weight [M, N] with rows ld_dw >= N elements apart (round 5: a column block of a wider gradient tensor -- the [H, n] halves of the
 *   visual encoder's [H, 2n] weights, box_rel.py:21-27 -- is written in place), d_bias [M] or NULL, x_colsum [N] or NULL in out_dtype (VLG_F32, or VLG_BF16 = the parameter's storage type: no cast
 *   launch behind the reduction; accumulation is fp32 either way); x_colsum = sum_rows x (the bias gradient when the roles
 *   are swapped: a weight stored [in, out] as in `matmul(child + parent, arc_encoder_w2) + arc_encoder_b`, joint.py:285-286,
 *   takes dy := the layer input and x := the cotangent).  ws: vlg_linear_wgrad_workspace(K, M, N) bytes (0 = unsupported shape).
 * Split over the token rows across the whole chip (bf16 MFMA, fp32 accumulate); partial tiles are added in a fixed
 * order: bit-reproducible, no atomics.
 *   in_dtype (round 5): VLG_BF16 as above, or VLG_F32 (the reference's `precision: 32`, config/trainer/train.yaml:20): dy and x are float32
 *   and every product a b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi on the bf16 matrix cores (hi = bf16(v), lo = bf16(v - hi), split
 *   between the staging registers and LDS): relative error <= ~2^-16 per product before the fp32 accumulation, operands read once. */
size_t vlg_linear_wgrad_workspace(int K, int M, int N);
int vlg_linear_wgrad(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, int in_dtype, void* ws, size_t ws_bytes, int out_dtype,
                     void* d_weight, int ld_dw, void* d_bias, void* x_colsum, void* stream);
/* The same in two steps, for a caller that issues SEVERAL weight gradients and needs none of them before its end (the parser's feed-forwards,
 * src/model/ldndmv.py:174-183 under loss.backward(): seven products): vlg_linear_wgrad_partial is the split-K launch alone (the partial tiles stay
 * in `ws`, which must stay alive and untouched; want_bias / want_x_colsum: which column sums to carry), vlg_linear_wgrad_reduce_group adds the
 * partial tiles of up to 12 such products per launch -- the same fixed-order sums, the same bits as vlg_linear_wgrad. */
typedef struct VlgWgradReduce {
    const void* ws;                      /* the workspace vlg_linear_wgrad_partial filled for (K, M, N) */
    void *d_weight, *d_bias, *x_colsum;  /* as in vlg_linear_wgrad (d_bias / x_colsum NULL when not carried) */
    int K, M, N, ld_dw, out_dtype;
    int in_dtype;                        /* the operand type the partial launch was given (it decides the split plan) */
} VlgWgradReduce;
int vlg_linear_wgrad_partial(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, int in_dtype, void* ws, size_t ws_bytes,
                             int want_bias, int want_x_colsum, void* stream);
int vlg_linear_wgrad_reduce_group(const VlgWgradReduce* items, int count, void* stream);
/* The split-K launches of SEVERAL products as one grid per kernel image (round 6): what vlg_linear_wgrad_partial does for each item, in up to four
 * launches for any number of bf16 items (a product is 256 workgroups -- one per CU -- and alone pays the chip's fill and drain; in one grid the next
 * product's workgroups start as the previous one's finish).  Same partial tiles, same bits; float32 items are launched one by one.  Every item's
 * operands and workspace must stay alive and untouched until vlg_linear_wgrad_reduce_group has run. */
typedef struct VlgWgradPartial {
    const void *dy, *x;                  /* [K, M] (rows ld_dy apart), [K, N] (ld_x) */
    void* ws;
    size_t ws_bytes;
    int ld_dy, ld_x, K, M, N, in_dtype, want_bias, want_x_colsum;
} VlgWgradPartial;
int vlg_linear_wgrad_partial_group(const VlgWgradPartial* items, int count, void* stream);

/* The two TRAINABLE encoders between the frozen features and the structured step (round 5; BASELINE.json configs[4]) -- what
 * `JointModelBase.forward` runs first, src/model/base.py:229,235.  The GEMMs are the caller's (library); these are the passes around them.
 *
 * vlg_dropout: nn.Dropout of `MLPEncoder.forward` (src/model/text_encoder/mlp_encoder.py:36-38) on the [rows, cols] embeddings, and its
 *   adjoint (the same call on the cotangent).  out = x * m (+ add), m = 0 or 1/(1-p):
 *     mask != NULL: explicit fp32 values [rows, cols] -- or, shared_rows > 0, one [cols] row per `shared_rows` consecutive rows
 *                   (SharedDropout, nn/dropout.py:42-63, `shared_dropout` of the encoder's config);
 *     rng  != NULL: a counter-based draw, Philox4x32-10 keyed by the DEVICE-resident pair rng[0] = seed, rng[1] = step over the element
 *                   index (`site` is added to the seed: one state serves every dropout layer of a step with independent streams) -- nothing is stored, the adjoint regenerates the same bits, and a HIP-graph replay sees the step that
 *                   vlg_rng_advance (a one-thread launch: rng[1] += 1) left there.  p is applied in steps of 2^-16.
 *   x (`dtype`), add / out (`out_dtype`; add NULL = none: the other producer of a gradient that has two), cols a multiple of 8,
 *   buffers 16-byte aligned.
 * vlg_vis_encoder: `VisBoxRelSimpleEncoder.forward` (src/model/vis_encoder/box_rel.py:29-52, img_feat: inputs = [box ; mean_r box]) and
 *   the concatenation of `vis_feat_unprune` (src/model/joint.py:137-171) behind its Linear layers: by linearity the caller computes
 *   P [B R, ldp] = X W_a^T (X [B R, n] the region features, W_a the first n columns of the F stacked [H, 2n] weights) and
 *   C [B, ldp] = mean_r(X) W_b^T + bias; this writes mid [B, V, H] (`dtype`):
 *     rows off_box + r          LeakyReLU(P[b r, col_box..] + C[b, col_box..])                       box_fc          (:47)
 *     rows off_rel + i R + j    LeakyReLU((P[b i, col_rel..] + P[b j, col_rel..]) / 2 + C[b, col_rel..])   rel_fc   (:41-45; 657 GFLOP as written)
 *     rows off_attr + r         as box, attr_fc                                                       (:48-49)
 *     row  off_img              mean_r of the box rows (all R of them, as joint.py:163 does)          add_image
 *   (-1 for an absent factor; the present encoders' column blocks are adjacent from column 0; H/4 divides 256; ldp a multiple of 8).
 * vlg_vis_encoder_backward: grad_mid [B, V, H] -> dP [B R, ldp] and dC [B, ldp] = sum_r dP[b r] (the per-image term's cotangent; fixed order);
 *   the weight gradients are then two vlg_linear_wgrad calls, dP^T X into columns [0, n) and dC^T mean_r(X) into [n, 2n) of the stack. */
int vlg_dropout(const void* x, const float* mask, int shared_rows, const uint64_t* rng, unsigned site, float p, const void* add, void* out, long long rows,
                int cols, int dtype, int out_dtype, void* stream);
int vlg_rng_advance(uint64_t* rng, void* stream);
/* out[i] = 0 or 1/(1-p), i < n: the same draw written out as explicit fp32 masks -- for the SMALL dropout layers of a step (the SharedDropout
 * rows [B, d] of the word / child / parent encoders and of head_ff, nn/dropout.py:42-63), all of them in one launch. */
int vlg_dropout_mask(const uint64_t* rng, unsigned site, float p, float* out, long long n, void* stream);
int vlg_vis_encoder(const void* P, const void* C, int B, int R, int H, int V, int ldp, int col_box, int col_rel, int col_attr, int off_box, int off_rel,
                    int off_attr, int off_img, int dtype, float slope, void* mid, void* stream);
int vlg_vis_encoder_backward(const void* P, const void* C, const void* grad_mid, int B, int R, int H, int V, int ldp, int col_box, int col_rel, int col_attr,
                             int off_box, int off_rel, int off_attr, int off_img, int dtype, float slope, void* dP, void* dC, void* stream);

/* The byte work of `lang_feat_max_tree` (src/model/joint.py:235-292) between the DP, the encoder GEMMs and the arc encoder.
 * Shapes: B sentences, L words, N = L + 1 positions (root first), h encoder width, d matching width; M = B*N rows.
 * Activations are stored as bf16 or fp32 (`act_dtype` / `out_dtype`; fp32 is the reference's `precision: 32`,
 * config/trainer/train.yaml:20); arithmetic is fp32 either way.  `drop` [B,3,d] fp32 or NULL: the SharedDropout masks
 * (nn/dropout.py:42-63; 0 or 1/(1-p), shared over the positions of a sentence) of the word | child | parent encoders, applied
 * AFTER the activation as `MLP.forward` does (nn/common.py:47-51); NULL = identity (eval mode); `ld_drop` = elements between
 * consecutive sentences' masks (3d when contiguous; a [B,4,d] draw that also holds `lang_feat_word_only`'s mask passes 4d).
 *   root_cat          x [B,L,h] (in_dtype) -> x1 [B,N,h] (out_dtype): row 0 = masked mean of the words (joint.py:262-265), rows 1.. = x (:266)
 *   root_cat_backward d_x1 [B,N,h] (in_dtype) -> d_x [B,L,h] fp32
 *   split             pre [M,3d] = x1 W_cat^T + b_cat (word | child | parent encoders, joint.py:267-273) ->
 *                     txt[b,n,:] = word third (txt is [B,2N,d], the word half, :288), child [M,d] = LeakyReLU(child third),
 *                     parent [M,d] = LeakyReLU(parent third of row heads[b,n]) (gather by the predicted heads, :271-273),
 *                     each times its dropout mask; sum [M,d] = child + parent (optional; the operand of the affine term, :285)
 *   split_backward    d_txt [B,2N,d] (dtype; the word half is read), d_child / d_parent fp32 [M,d], d_sum [M,d] (d_sum_dtype) or NULL
 *                     (added to both), child / parent (activations) -> d_pre [M,3d] (dropout mask, LeakyReLU', scatter-add by head
 *                     in ascending row order)
 *   marginal          grad_attach [B,N,N,2] fp32, heads [B,N], lengths [B] -> txt_marginal [B,2N] fp32 = cat([mask,
 *                     arc_margin.gather(-1, predicted)]) (joint.py:246-261; use_marginal 0: cat([mask, mask]), :262),
 *                     txt_mask [B,2N] u8 = cat([mask, mask]) with the root slot masked (:248-249)
 *   arc_out           tri [M,d] fp32 (+ aff [M,d] act_dtype or NULL) -> txt[b, N+n, :] (arc_repr, joint.py:278-288)
 *   rowscale          out[b,n,c] = pre[b,n,c] * drop[b*ld_drop + c] for c < d, 0 for d <= c < width  (one encoder's SharedDropout on [B,N,d]: the word
 *                     encoder of `lang_feat_word_only`, joint.py:193-211, and its adjoint; drop NULL = identity; rows ld_in / ld_out elements apart
 *                     (round 5): the word third of the three encoders' SHARED projection [B N, 3d] is read in place, and the adjoint writes the
 *                     full-width cotangent of that projection, zeros for the other thirds, in one pass; in place allowed with equal strides) */
int vlg_langfeat_root_cat(const void* x, const int64_t* lengths, int B, int L, int h, int in_dtype, void* x1, int out_dtype,
                          void* stream);
int vlg_langfeat_root_cat_backward(const void* d_x1, const int64_t* lengths, int B, int L, int h, int in_dtype, void* d_x, int out_dtype,
                                   void* stream);
int vlg_langfeat_split(const void* pre, const int64_t* heads, const float* drop, int ld_drop, int B, int N, int d, int act_dtype, float slope,
                       void* txt, void* child, void* parent, void* sum, void* stream);
int vlg_langfeat_split_backward(const void* d_txt, int d_txt_dtype, const float* d_child, const float* d_parent, const void* d_sum,
                                int d_sum_dtype, const void* child, const void* parent, const int64_t* heads, const float* drop, int ld_drop,
                                int B, int N, int d, int act_dtype, float slope, void* d_pre, void* stream);
int vlg_langfeat_marginal(const float* grad_attach, const int64_t* heads, const int64_t* lengths, int B, int N, int use_marginal,
                          float* txt_marginal, uint8_t* txt_mask, void* stream);
int vlg_langfeat_arc_out(const float* tri, const void* aff, int B, int N, int d, int act_dtype, void* txt, void* stream);
int vlg_langfeat_rowscale(const void* pre, int ld_in, const float* drop, int B, int N, int d, int ld_drop, int act_dtype, void* out, int ld_out, int width,
                          void* stream);

/* Small (batched) matrix products in weight space -- the folded bottleneck weights W1 W0 of `DMVSkipConnectEncoder`
 * (src/model/nn/dmv_spec.py:52-54) and their unfolding, the per-sentence context term of `head_ff` (src/model/ldndmv.py:174-177): products
 * whose OUTPUT is a few hundred rows and columns, which a library GEMM maps to one workgroup.  One wavefront per 32 x 32 tile per batch entry.
 *   C[z][m][n] = alpha * sum_k A[z](m,k) B[z](k,n) + bias[z][n] + u[z][m] v[z][n] (+ C[z][m][n] if accumulate)
 *   A(m,k) at a + z sab + m sam + k sak,  B(k,n) at b + z sbb + k sbk + n sbn  (ELEMENT strides: transposed / sliced operands in place);
 *   C row-major, rows ldc apart, batch entries scb apart; bias [N] (batch stride sbias), u [M] (su), v [N] (sv): in_dtype, any may be NULL
 *   (u and v together).  in_dtype VLG_BF16 (bf16 products) or VLG_F32 (exact fp32 products), fp32 accumulation, out_dtype either. */
int vlg_small_gemm(const void* a, long long sab, long long sam, long long sak, const void* b, long long sbb, long long sbk, long long sbn,
                   void* c, long long scb, long long ldc, const void* bias, long long sbias, const void* u, long long su, const void* v,
                   long long sv, int batch, int M, int N, int K, float alpha, int accumulate, int in_dtype, int out_dtype, void* stream);

/* Several INDEPENDENT small products as ONE launch (round 5): one dependency level of the weight-space products of the parser's
 * feed-forwards (src/model/ldndmv.py:174-183, nn/dmv_spec.py:38-54) -- each field as the vlg_small_gemm argument of the same name.
 * Problems may differ in shape, strides and dtypes; none may read what another one of the same call writes. */
typedef struct VlgSmallGemm {
    const void *a, *b, *bias, *u, *v;
    void* c;
    long long sab, sam, sak, sbb, sbk, sbn, scb, ldc, sbias, su, sv;
    int batch, M, N, K, accumulate, in_dtype, out_dtype;
    float alpha;
} VlgSmallGemm;
int vlg_small_gemm_group(const VlgSmallGemm* problems, int count, void* stream);

/* Element-wise passes between the library GEMMs of the parser's feed-forwards (vlgae_amd/parser_ff.py): `MLP`
 * (src/model/nn/common.py:23-51: Linear -> LeakyReLU -> SharedDropout) and `DMVSkipConnectEncoder` (src/model/nn/dmv_spec.py:38-54).
 * Activations in act_dtype (VLG_BF16 / VLG_F32), rows of H channels (H a multiple of 8), 16-byte aligned; fp32 arithmetic.
 *   vlg_ff_context_mean       out [B,h] (out_dtype) = mean over ALL L positions of x [B,L,h] (in_dtype) read as out_dtype values: `extract_sent_repr`
 *                             of context_mode 'mean' (src/model/ldndmv.py:226), cast + reduction in one launch.
 *   vlg_ff_mlp_act            x [B L + Ms, H] in place.  Rows < B L: LeakyReLU(x + cterm[row / L]) * drop_head[row / L] (cterm [B,H]
 *                             act_dtype: the sentence's context columns + bias; drop_head [B,H] fp32 or NULL); the Ms rows behind
 *                             them (2-D inputs of their MLPs): LeakyReLU(x) * drop_small[row - B L] (fp32 [Ms] or NULL).
 *   vlg_ff_act                out[m,j'] = LeakyReLU(in[m,j] + residual[m]) * mask[m,j'] * mask_scale, in [M,J,H], residual [M,H] or NULL, mask
 *                             (act_dtype, indexed like out; e.g. an nn.Dropout keep-mask of 0 / 1 with mask_scale = 1 / (1 - p)) or NULL.  swap = 0: j' = j, out may be in.  swap = 1 (J = 4): in is
 *                             [m,val,dir], out [m,dir,val] -- the stack of nn/dmv_spec.py:47 as a store permutation.
 *                             rng != NULL (then mask = NULL, swap = 0): the keep-mask is the counter-based draw of vlg_dropout (site, p) over the output's
 *                             element index and mask_scale is 1 / (1 - p) -- mid_ff's nn.Dropout (nn/dmv_spec.py:52) without its 21 MB mask tensor.
 *   vlg_ff_act_backward       out[m,j'] = LeakyReLU'(act[m,j]) * g[m,j] * mask[m,j] * mask_scale (g, act, mask [M,J,H] in the same order; the
 *                             derivative from the sign of the stored activation); sum [M,H] fp32 (or NULL) = / += (accumulate)
 *                             sum_j of the stored out values.  swap as above (g, act in [m,dir,val]; out in [m,val,dir]).
 *   vlg_ff_mlp_act_backward   gpre = LeakyReLU'(x) * mask * (gx + t): gx fp32, t act_dtype or NULL, masks as in vlg_ff_mlp_act. */
int vlg_ff_context_mean(const void* x, int in_dtype, int B, int L, int h, void* out, int out_dtype, void* stream);
int vlg_ff_mlp_act(void* x, const void* cterm, const float* drop_head, const float* drop_small, int B, int L, int Ms, int H, int act_dtype,
                   float slope, void* stream);
int vlg_ff_act(const void* in, const void* residual, const void* mask, float mask_scale, const uint64_t* rng, unsigned site, float p, void* out,
               long long M, int J, int H, int swap, int act_dtype, float slope, void* stream);
int vlg_ff_act_backward(const void* g, const void* act, const void* mask, float mask_scale, const uint64_t* rng, unsigned site, float p, void* out,
                        float* sum, long long M, int J, int H, int swap, int accumulate, int act_dtype, float slope, void* stream);
int vlg_ff_mlp_act_backward(const float* gx, const void* t, const void* x, const float* drop_head, const float* drop_small, void* gpre, int B, int L,
                            int Ms, int H, int act_dtype, float slope, void* stream);
/* The root rule and the cotangent of the small projection product (round 5; src/model/ldndmv.py:205 `root_scorer(h_root, h_child).sum([-1,-2])
 * .log_softmax(-1)`).  small [4 (T + 3), ld] (act_dtype) holds, for row (token c, (dir,val) dv) = 4c + dv, the column blocks attach.project2 |
 * root.project2 | . | . and, in rows 4T + dv, block 2 = root.project1 of the root's representation; r = the scorers' rank.
 *   vlg_ff_root_rule           root_rule [T] fp32 = log_softmax_c sum_{dv,e} r1[dv,e] r2[c,dv,e]
 *   vlg_ff_root_rule_backward  g_small [4 (T + 3), 4r] (act_dtype, contiguous) = the whole cotangent of `small`: block 0 <- g_x2 [4T, r] (rows ld_x2 apart;
 *                              NULL = 0), block 1 / 2 <- the root rule's adjoint from g_root [T] fp32, block 3 of the last 8 rows <- g_y2 [8, r] (ld_y2),
 *                              zeros elsewhere: one pass instead of a zero-fill, two copies, a softmax adjoint and two products. */
int vlg_ff_root_rule(const void* small, int ld, int T, int r, int act_dtype, float* root_rule, void* stream);
int vlg_ff_root_rule_backward(const void* small, int ld, int T, int r, int act_dtype, const float* root_rule, const float* g_root, const void* g_x2,
                              int ld_x2, const void* g_y2, int ld_y2, void* g_small, void* stream);
/* A feed-forward Linear over all token rows FUSED with the element-wise pass behind it (round 6; the DMVSkipConnectEncoder stages of
 * src/model/nn/dmv_spec.py:38-54): one row-streaming launch in place of a library GEMM + vlg_ff_act / vlg_ff_act_backward pair.  bf16 storage,
 * exactly 256 output channels per column block and 256 input channels (the backward launch also takes 512 and 32); every pointer 16-byte aligned, rows of x / g `ld` elements apart.
 *   vlg_ff_linear_act           x [rows, 256], w [nb 256, 256] (nn.Linear layout: w[n][k]), bias [nb 256] or NULL, nb = 1 or 2 column blocks y:
 *                               out[orow][c] = LeakyReLU(bf16(x[row] . w[256 y + c] + bias) + residual[row >> rs][c]) * keep[orow][c],
 *                               orow = (row >> rs) om + y oy + (row & ((1 << rs) - 1)); residual [rows >> rs, 256] or NULL.
 *                                 plain layer: nb 1, rs 0, om 1, oy 0;  the (no | has) bottlenecks with their skip connection: nb 2, rs 0, om 2, oy 1,
 *                                 residual = x;  the (left | right) bottlenecks: rows (m,val), nb 2, rs 1, om 4, oy 2 -> out [m,dir,val] (the stack of :47).
 *                               keep: mask [out rows, 256] bf16 times mask_scale, or -- rng -- the counter-based draw of (site, p) over the output's
 *                               element index, as vlg_ff_act; both NULL: none.
 *   vlg_ff_linear_act_backward  g [rows, 256], w_t [256, 256] = the layer's weight TRANSPOSED (w_t[n][k] = weight[k][n], so that g . w_t[n] is the
 *                               cotangent of input channel n): out[orow][c] = LeakyReLU'(act[row][c]) * bf16(g[row] . w_t[c]) * keep[row][c];
 *                               rows are groups m J + j (J = 1, 2 or 4), sum [rows / J, 256] fp32 (or NULL) = / += (accumulate) sum_j of the stored
 *                               out values; swap (J = 4): orow = 4 m + (dir,val <- val,dir)(j) -- as vlg_ff_act_backward on the product.
 *                               k = the contraction length (columns of g): 256 as above; 512: g [rows, 512] is the cotangent of a two-block stage
 *                               and w_t [2, 256, 256] the transposes of the weight's two [256, 256] row blocks one after the other (vlg_ff_transpose256
 *                               of the blocks); 32 with w_kn = 1: g [rows, 32] and w_t is the weight ITSELF, [32, 256] (w[k][n]: the folded
 *                               projections' cotangent, 2 r = 32 columns).  w_kn = 0 otherwise. */
int vlg_ff_linear_act(const void* x, int ldx, const void* w, const void* bias, long long rows, int nb, const void* residual, int rs, int om, int oy,
                      const void* mask, float mask_scale, const uint64_t* rng, unsigned site, float p, void* out, float slope, void* stream);
int vlg_ff_linear_act_backward(const void* g, int ldg, const void* w_t, int k, int w_kn, long long rows, int J, const void* act, const void* mask, float mask_scale,
                               const uint64_t* rng, unsigned site, float p, void* out, float* sum, int swap, int accumulate, float slope,
                               void* stream);
/* TWO consecutive 256 -> 256 stages in one launch (round 6): stage 2 runs on stage 1's stored rows without their round trip through memory (stage 1's
 * epilogue leaves its tile in LDS as well; both weight blocks sit in registers).  Each stage is what vlg_ff_linear_act (backward = 0: w in nn.Linear
 * layout, bias) / vlg_ff_linear_act_backward at k = 256 (backward = 1: w = the transposed weight, act, J / sum / swap / accumulate) computes with these
 * arguments; stage 1 keeps its rows (a plain layer: one column block, no skip connection; backward: J = 1, no sum, no permutation) and still writes
 * its own `out` (the weight gradients read it), stage 2 forward is a plain layer too.  The parser's direction -> output stages (nn/dmv_spec.py:52-54)
 * and their adjoints.  mask / rng / site / p / mask_scale as in vlg_ff_linear_act; unused fields zero. */
typedef struct VlgFfStage {
    const void *w, *bias;                /* [256, 256] bf16; bias [256] or NULL (forward) */
    const void* mask;                    /* keep-mask [rows, 256] bf16 or NULL */
    const uint64_t* rng;                 /* or the counter-based draw */
    void* out;                           /* [rows, 256] bf16 */
    const void* act;                     /* backward: the stored activations [rows, 256] */
    float* sum;                          /* backward: [rows / J, 256] fp32 or NULL */
    float mask_scale, p;
    unsigned site;
    int J, swap, accumulate;
} VlgFfStage;
int vlg_ff_linear_act_chain2(const void* x, int ldx, long long rows, int backward, const VlgFfStage* s1, const VlgFfStage* s2, float slope, void* stream);
/* out[rows, ncols] (bf16, rows ldo elements apart) = bf16(x[rows, 256] @ w[256, ncols]): a Linear's INPUT gradient (g @ weight, weight [256, in] as nn.Linear
 * stores it) or any product of token rows with a weight whose contraction index is its slow one, as one row-streaming launch: the weight is read
 * where it lies (rows ldw elements apart: a column slice of a wider matrix is fine), 256 output columns per workgroup column block.  ncols a multiple
 * of 8, ldo of 4; x / out 16-byte aligned.  rng (or NULL): the result times the counter-based keep-mask of (site, p) drawn over out's [rows, ncols]
 * element index exactly as vlg_dropout draws it -- the adjoint of Linear(Dropout(x)) in one launch.  (MLP.linear / MLPEncoder.linear under loss.backward(): src/model/nn/common.py:30, text_encoder/mlp_encoder.py:36-40.) */
int vlg_ff_linear_kn(const void* x, int ldx, const void* w, int ldw, long long rows, int ncols, const uint64_t* rng, unsigned site, float p, void* out, int ldo,
                     void* stream);
/* The head of the skip-connect encoder's adjoint in one launch (vlg_ff_linear_act_backward at k = 512 with vlg_ff_mlp_act_backward's element-wise pass):
 * out[row][c] = LeakyReLU'(x[row][c]) * keep * (add[row][c] + bf16(g[row] . w_t[c])), g [rows, 512] bf16, w_t [2, 256, 256] as above, add [rows, 256] fp32
 * (the skip connections' cotangent), x [rows, 256] the stored MLP outputs; keep = drop_head[row / L] (fp32 [M0 / L, 256], one SharedDropout mask per
 * sentence) for rows < M0 and drop_small[row - M0] (fp32, one value per row) behind them, either NULL = none (src/model/nn/common.py:47-51). */
int vlg_ff_linear_mlp_act_backward(const void* g, int ldg, const void* w_t, long long rows, const float* add, const void* x, const float* drop_head,
                                   const float* drop_small, long long M0, int L, void* out, float slope, void* stream);
/* out [n, 256, 256] (bf16): out[z][j][k] = w[z][k][j] for the n <= 8 contiguous 256 x 256 bf16 matrices w[0..n-1] (a HOST array of n device
 * pointers) -- the `w_t` operands of the backward launches above in one launch. */
int vlg_ff_transpose256(const void* const* w, int n, void* out, void* stream);

/* Score construction feeding the DP -- the tensor half of `DiscriminativeNDMV._forward`, src/model/ldndmv.py:179-209 with the
 * factorised-bilinear scorers of src/model/nn/dmv_spec.py:57-76: from the scorers' projected inputs to the root-merged
 * potentials, without the [B,L,T,2,2] rule table.
 *   x1 [B,L,2,2,r] = attach_scorer.project1(h_parent), x2 [T,2,2,r] = attach_scorer.project2(h_child)      (in_dtype)
 *   y1 [B,L,2,2,r] = dec_scorer.project1(h_parent),    y2 [2,2,2,r] = dec_scorer.project2(h_dec)           (in_dtype)
 *   root_rule [T] fp32 = the root scorer's log-softmax over tokens (batch-independent, ldndmv.py:205)
 *   token [B,L] int64 in [0,T), head_mask [B,L] u8 or NULL (function-word heads, ldndmv.py:195-199), mask_fill = -INF of
 *   src/__init__.py:110.   Outputs (out_dtype): merged_dec [B,L+1,2,2,2], merged_attach [B,L+1,L+1,2] exactly as
 *   `DMV1o.merge(dec, attach, root)` lays them out (distributions.py:253-265; zero = -1e12, one = 0).
 *   x1, x2, y1, y2 are rows of r values (one per (position | token | decision, direction, valence)) ld_* ELEMENTS apart: ld = r
 *   is a contiguous tensor, a wider stride takes a column slice of a GEMM output that holds several projections side by side
 *   (vlgae_amd/parser_ff.py computes attach.project1 | dec.project1 in one product) without a copy.
 * backward: cotangents of the two merged tensors (fp32; e.g. the DP's expected counts) -> d_x1, d_y1 [B,L,2,2,r] (rows ld_dx1 /
 *   ld_dy1 apart), d_x2 [T,2,2,r], d_y2 [2,2,2,r] (contiguous) in grad_dtype (VLG_F32 / VLG_BF16: the storage type of the layer
 *   that receives them), d_root_rule [T] fp32.  The softmax weights are recomputed; the batch-shared tables' gradients are
 *   per-sentence fp32 partials added in sentence order (ws: vlg_ndmv_potentials_backward_workspace bytes).
 * One workgroup per sentence with the token table in LDS: L, T, r must fit 160 KiB (VLG_ERR_SHAPE otherwise). */
int vlg_ndmv_potentials(const void* x1, int ld_x1, const void* x2, int ld_x2, const void* y1, int ld_y1, const void* y2, int ld_y2,
                        const float* root_rule, const int64_t* token, const uint8_t* head_mask, int B, int L, int T, int r, int in_dtype,
                        float mask_fill, int out_dtype, void* merged_dec, void* merged_attach, void* stream);
size_t vlg_ndmv_potentials_backward_workspace(int B, int L, int T, int r);
int vlg_ndmv_potentials_backward(const void* x1, int ld_x1, const void* x2, int ld_x2, const void* y1, int ld_y1, const void* y2, int ld_y2,
                                 const int64_t* token, const uint8_t* head_mask, const float* g_merged_dec, const float* g_merged_attach,
                                 int B, int L, int T, int r, int in_dtype, void* ws, size_t ws_bytes, int grad_dtype, void* d_x1, int ld_dx1,
                                 void* d_x2, void* d_y1, int ld_dy1, void* d_y2, float* d_root_rule, void* stream);

/* Viterbi pass with every output of the Max semiring in ONE launch: best score, the 0/1 counts of the best tree (what
 * `-DMV1o(...).max.sum()` back-propagates, ldndmv.py:277-281) and its head vector (`argmax`, joint.py:256-258) -- the two
 * call sites see the same potentials within a training step.  Any of grad_dec / grad_attach / heads may be NULL (not all). */
int vlg_dmv1o_viterbi(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype,
                      const float* grad_best, float* best_score, float* grad_dec, float* grad_attach, int64_t* heads,
                      void* ws, size_t ws_bytes, void* stream);

/* Marginals AND the Viterbi tree of the same potentials in one launch -- the pair `lang_feat_max_tree` takes every step
 * (src/model/joint.py:251-258), plus, for the parser's Viterbi-tree loss (src/model/ldndmv.py:277-281), the tree counts: grid (B, 2), the
 * two workgroups of a sentence (Log inside-outside | Max inside + back-pointer walk) share a CU as they do on two streams, without
 * the fork / join between queues.  Outputs as vlg_dmv1o_inside_outside (logZ [B], grad_dec [B,N,2,2,2] or NULL, grad_attach
 * [B,N,N,2]: unit upstream gradient) and vlg_dmv1o_viterbi (best_score [B], tree_dec / tree_attach or NULL, heads [B,N]); no
 * workspace.  Only where both passes keep everything in LDS and fit one CU together (vlg_dmv1o_marginals_viterbi_supported(N) == 1:
 * N <= 44); VLG_ERR_SHAPE otherwise -- launch the two entry points on two streams then. */
int vlg_dmv1o_marginals_viterbi_supported(int N);
int vlg_dmv1o_marginals_viterbi(const void* dec, const void* attach, const int64_t* lengths, int B, int N, int in_dtype, float* logZ,
                                float* grad_dec, float* grad_attach, float* best_score, float* tree_dec, float* tree_attach, int64_t* heads,
                                void* stream);

/* Chain rule on the API path (helpers.py:116-157: autograd scales the unit-upstream counts by d loss / d logZ):
 *   out_a[b,:] = counts_a[b,:] * g[b*g_stride], out_b likewise; counts fp32, outputs out_dtype (VLG_F32 / VLG_BF16, round to
 *   nearest even like torch's cast).  g_stride 1: g [B]; 0: one scalar (the expanded gradient of `.sum()`).  n_a / n_b =
 *   elements per sentence; either side may be empty (n = 0, pointers then ignored). */
int vlg_scale_counts(const float* counts_a, const float* counts_b, const float* g, int g_stride, int B, int n_a, int n_b,
                     int out_dtype, void* out_a, void* out_b, void* stream);

/* ---- Data feed (host code; no device work, no stream).  SURVEY.md section 8 row f4. ----
 *
 * Length bucketing -- ConstantTokenNumSampler.kmeans, src/datamodule/sampler.py:148-191: Lloyd iterations on the sentence
 * lengths from the caller's initial centroids (the reference draws them with torch.randperm over the distinct lengths; the
 * host mirror draws them the same way so that the buckets are identical).  Same tie rules as the reference: nearest centroid
 * with the lowest id winning ties; an empty cluster takes the farthest point of the biggest cluster.
 *   seq_len [n] int32, init_centroids [k], 1 <= k <= n.  centroids [k] (first *n_clusters valid), assign [n] = bucket id in
 *   0..*n_clusters-1 (empty clusters dropped, order kept).  Centroids are exact while a bucket holds < 2^24 tokens. */
int vlg_feed_kmeans(const int32_t* seq_len, int64_t n, const float* init_centroids, int k, int max_it, float* centroids,
                    int32_t* assign, int* n_clusters);

/* One epoch's batches -- ConstantTokenNumSampler._init_iter + _process_batch, sampler.py:86-140.
 *   bucket_offsets [n_buckets+1], bucket_items (CSR: sentence ids per bucket), chunks [n_buckets] = batches per bucket,
 *   bucket_perms = the per-bucket permutations concatenated (positions within the bucket), batch_perm [sum(chunks)] = order
 *   of the raw batches.  single_sent_threshold: sentences at least this long become batches of one (-1 = off);
 *   sort_in_batch: stable sort by decreasing length.
 *   out_offsets [sum(chunks) + n + 1], out_items [n]: CSR of the epoch's batches, *n_batches of them. */
int vlg_feed_batches(const int32_t* seq_len, int64_t n, const int64_t* bucket_offsets, const int64_t* bucket_items, int n_buckets,
                     const int64_t* chunks, const int64_t* bucket_perms, const int64_t* batch_perm, int single_sent_threshold,
                     int sort_in_batch, int64_t* out_offsets, int64_t* out_items, int64_t* n_batches);

/* Region-feature collate -- _COCODetFeatLazyLoader.__call__, src/datamodule/task/vlparse.py:36-92.
 * vlg_feed_npy_shape: rows / columns of a 2-d little-endian f2 / f4 / f8 C-order .npy.
 * vlg_feed_collate_npy: image i keeps n_sel[i] rows of paths[i] -- rows sel[i*sel_stride + 0..] (sel NULL = the leading rows) --
 *   split into [feat_dim | box_dim] columns and written as float32 into feat [n, max_len, feat_dim], box [n, max_len, box_dim],
 *   mask [n, max_len] (1 = real row); rows beyond n_sel[i] are zero.  Every byte of the three outputs is written, so they may
 *   be uninitialised (pinned) staging memory.  n_threads reader threads (pread; no shared state). */
int vlg_feed_npy_shape(const char* path, int64_t* rows, int64_t* cols);
int vlg_feed_collate_npy(const char* const* paths, int n, const int32_t* sel, int sel_stride, const int32_t* n_sel, int feat_dim,
                         int box_dim, int max_len, float* feat, float* box, uint8_t* mask, int n_threads);

/* Device self-test of the cross-lane (DPP / ds_swizzle) exchange primitives the DP kernels rely on.
 * `scratch` = one device int; after the stream drains it holds 0 iff the primitives behave as assumed. */
int vlg_selftest_xlane(int* scratch, void* stream);

/* Thread-local message for the last non-zero return on this thread ("" if none). */
const char* vlg_last_error(void);

/* Library / ABI version, e.g. 142 = 0.1.4.2 (round 6, 142: vlg_attn_fuse / vlg_attn_fuse_backward take key_chunk and a workspace -- the key-split form for the
 * shipped 1369-key factor layout -- and the gradients' storage type; vlg_attn_fuse_workspace added; round 5, 141: vlg_dropout, vlg_rng_advance, vlg_vis_encoder(_backward) added, vlg_linear_wgrad takes ld_dw and in_dtype;
 * the Python binding refuses a library whose version differs from the one it was written against; round 4: vlg_langfeat_* take the activations' storage type and the SharedDropout masks,
 * vlg_langfeat_rowscale, vlg_ff_* added, vlg_ndmv_potentials* take row strides and the gradients' storage type; round 3, 120: vlg_linear_wgrad, vlg_langfeat_*, vlg_ndmv_potentials*, vlg_dmv1o_viterbi added;
 * round 2, 110: vlg_bilinear_align_backward takes a workspace; vlg_scale_counts, vlg_feed_* added). */
int vlg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* VLGAE_AMD_H */
