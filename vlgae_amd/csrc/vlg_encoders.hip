// vlg_encoders.hip -- the two TRAINABLE encoders between the frozen features and the structured step (gfx950), BASELINE.json
// configs[4] "frozen BERT + Faster-RCNN feats -> ...": what `JointModelBase.forward` runs first, src/model/base.py:229,235.
//
//   MLPEncoder.forward            src/model/text_encoder/mlp_encoder.py:36-40   x = Linear_{800->256, no bias}(Dropout_p(emb))
//   VisBoxRelSimpleEncoder.forward src/model/vis_encoder/box_rel.py:29-52        inputs = [box ; mean_r box]; box_fc / rel_fc / attr_fc
//     + the concatenation of `vis_feat_unprune`, src/model/joint.py:137-171     (MLP = Linear -> LeakyReLU, nn/common.py:23-51)
//
// The GEMMs stay with the library (they have ~10^4 rows); what is here is the byte work around them, one pass each:
//   dropout_kernel          nn.Dropout on the [B L, 800] embeddings: explicit mask (tests: the reference's recorded draw) or a
//                           counter-based Philox4x32-10 draw keyed by a DEVICE-resident (seed, step) pair -- no 33 MB mask tensor, a
//                           HIP-graph replay draws fresh masks (the step counter is advanced on the device), and the backward pass
//                           regenerates the same bits instead of reading them back.
//   vis_encoder_fwd_kernel  the three MLPs' epilogues straight into the [B, V, H] factor tensor `vis_feat_unprune` concatenates:
//                           by linearity W [x_r ; m] = W_a x_r + W_b m, so P = X W_a^T is ONE [B R, n] x [n, F H] GEMM for the F
//                           encoders, C = mean_r(X) W_b^T + bias one row per image, and
//                             box / attr [b,r]   = LeakyReLU(P[b,r] + C[b])
//                             rel [b,i,j]        = LeakyReLU((P[b,i] + P[b,j]) / 2 + C[b])        (never the [B,R,R,4096] mean tensor)
//                             img [b]            = mean_r box[b,r]                                  (joint.py:162-171, add_image)
//   vis_encoder_bwd_kernel  the adjoint: d_mid [B,V,H] -> dP [B R, F H]; the per-image term's cotangent is its segment sum,
//   vis_segsum_kernel       dC[b] = sum_r dP[b,r] (pre = P_r + C for box / attr; for rel, s symmetric => sum_i dP_i = sum_ij s_ij g_ij).
// HBM-bound byte work: 16-byte accesses, fp32 arithmetic, one rounding per stored element; no matrix cores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"
#include "vlg_rng.h"
#include "vlg_rows.h"

namespace vlg {

namespace {

constexpr int kEncThreads = 256;

// out = x * mask (+ add): mask = explicit fp32 values (0 or 1/(1-p); [rows, cols] or, shared != 0, one [cols] row per `shared` rows) or
// the counter-based draw.  `add` (same storage type as out) is the other contribution of a gradient with two producers.
template <typename A, typename O>
__global__ __launch_bounds__(kEncThreads) void dropout_kernel(const A* __restrict__ x, const float* __restrict__ mask, int shared,
                                                              const uint64_t* __restrict__ rng, uint32_t site, uint32_t thr, float scale,
                                                              const O* add, O* out, size_t rows, int cols) {
    const int cv = cols >> 3;
    const size_t i = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    if (i >= rows * cv) return;
    const size_t row = i / cv;
    const int c = (int)(i - row * cv) * 8;
    float v[8], m[8];
    load8(x + row * cols + c, v);
    if (mask) load8(mask + (shared ? row / shared : row) * cols + c, m);
    else keep8(rng, site, i, thr, scale, m);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= m[k];
    if (add) {
        float t[8];
        load8(add + row * cols + c, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    store8(out + row * cols + c, v);
}

__global__ void rng_advance_kernel(uint64_t* rng) { rng[1] += 1; }

// explicit masks for the SMALL dropout layers of a step (SharedDropout's [B, d] rows, nn/dropout.py:42-63): out[i] = 0 or scale
__global__ __launch_bounds__(kEncThreads) void dropout_mask_kernel(const uint64_t* __restrict__ rng, uint32_t site, uint32_t thr, float scale,
                                                                   float* __restrict__ out, size_t n) {
    const size_t g = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    if (g * 8 >= n) return;
    float m[8];
    keep8(rng, site, g, thr, scale, m);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (g * 8 + k < n) out[g * 8 + k] = m[k];
}

// ---- visual encoder ----
struct VisArgs {
    const void *P, *C;        // [B R, ldp], [B, ldp] (storage type A)
    void* mid;                // [B, V, H]
    int R, H, V, ldp;
    int col_box, col_rel, col_attr;      // column offsets of the three encoders inside P / C (-1: absent)
    int off_box, off_rel, off_attr, off_img;   // row offsets of the factors inside mid (-1: absent)
    float slope;
};

// grid (tasks, B), tasks = [R rel rows i] + box + [attr]; every block sweeps R rows of H channels (4 channels per thread)
template <typename A>
__global__ __launch_bounds__(kEncThreads) void vis_encoder_fwd_kernel(VisArgs a) {
    __shared__ float4 red[kEncThreads];
    const A* P = (const A*)a.P;
    const A* C = (const A*)a.C;
    A* mid = (A*)a.mid;
    const int b = blockIdx.y, R = a.R, H = a.H, tpr = H >> 2;
    const int c4 = (threadIdx.x % tpr) * 4, jl = threadIdx.x / tpr, jstep = kEncThreads / tpr;
    const int n_rel = a.off_rel >= 0 ? R : 0;
    int task = blockIdx.x;
    const A* Pb = P + (size_t)b * R * a.ldp;
    A* mb = mid + (size_t)b * a.V * H;
    float q[4];
    if (task < n_rel) {                 // rel row i: out[b, off_rel + i R + j] = act((P_i + P_j) / 2 + C)
        const int i = task, col = a.col_rel;
        float pi[4], cc[4];
        load4(Pb + (size_t)i * a.ldp + col + c4, pi);
        load4(C + (size_t)b * a.ldp + col + c4, cc);
        A* ob = mb + ((size_t)a.off_rel + (size_t)i * R) * H;
        for (int j = jl; j < R; j += jstep) {
            load4(Pb + (size_t)j * a.ldp + col + c4, q);
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = leaky(0.5f * (pi[k] + q[k]) + cc[k], a.slope);
            store4(ob + (size_t)j * H + c4, q);
        }
        return;
    }
    task -= n_rel;
    const bool is_box = task == 0;
    const int col = is_box ? a.col_box : a.col_attr, off = is_box ? a.off_box : a.off_attr;
    float cc[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
    load4(C + (size_t)b * a.ldp + col + c4, cc);
    for (int r = jl; r < R; r += jstep) {
        load4(Pb + (size_t)r * a.ldp + col + c4, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            q[k] = leaky(q[k] + cc[k], a.slope);
            acc[k] += stored<A>(q[k]);
        }
        store4(mb + ((size_t)off + r) * H + c4, q);
    }
    if (is_box && a.off_img >= 0) {     // encoded["box"].mean(1, keepdim=True), joint.py:163 (all R rows, padded ones included, as there)
        red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (jl == 0) {
            for (int k = 1; k < jstep; ++k) {
                const float4 t = red[threadIdx.x + k * tpr];
                acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
            }
            const float inv = 1.f / (float)R;
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] *= inv;
            store4(mb + (size_t)a.off_img * H + c4, acc);
        }
    }
}

struct VisBwdArgs {
    const void *P, *C, *g;    // g: cotangent of mid [B, V, H]
    void* dP;                 // [B R, ldp]
    void* dC;                 // [B, ldp] written by the box / attr workgroups themselves, or null (a relation factor: vis_segsum_kernel follows)
    int R, H, V, ldp;
    int col_box, col_rel, col_attr, off_box, off_rel, off_attr, off_img;
    float slope;
};

template <typename A>
__global__ __launch_bounds__(kEncThreads) void vis_encoder_bwd_kernel(VisBwdArgs a) {
    __shared__ float4 red[kEncThreads];
    const A* P = (const A*)a.P;
    const A* C = (const A*)a.C;
    const A* g = (const A*)a.g;
    A* dP = (A*)a.dP;
    const int b = blockIdx.y, R = a.R, H = a.H, tpr = H >> 2;
    const int c4 = (threadIdx.x % tpr) * 4, jl = threadIdx.x / tpr, jstep = kEncThreads / tpr;
    const int n_rel = a.off_rel >= 0 ? R : 0;
    int task = blockIdx.x;
    const A* Pb = P + (size_t)b * R * a.ldp;
    const A* gb = g + (size_t)b * a.V * H;
    A* dPb = dP + (size_t)b * R * a.ldp;
    float q[4];
    if (task < n_rel) {   // dP_rel[b,i] = 1/2 sum_j s_ij (g[b,i,j] + g[b,j,i]), s = LeakyReLU'(pre), pre symmetric in (i, j)
        const int i = task, col = a.col_rel;
        float pi[4], cc[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
        load4(Pb + (size_t)i * a.ldp + col + c4, pi);
        load4(C + (size_t)b * a.ldp + col + c4, cc);
        const A* gr = gb + (size_t)a.off_rel * H;
        for (int j = jl; j < R; j += jstep) {
            float gij[4], gji[4];
            load4(Pb + (size_t)j * a.ldp + col + c4, q);
            load4(gr + ((size_t)i * R + j) * H + c4, gij);
            load4(gr + ((size_t)j * R + i) * H + c4, gji);
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] += (0.5f * (pi[k] + q[k]) + cc[k] > 0.f ? 1.f : a.slope) * (gij[k] + gji[k]);
        }
        red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (jl == 0) {    // fixed-order sum over the block's j-slices
            for (int k = 1; k < jstep; ++k) {
                const float4 t = red[threadIdx.x + k * tpr];
                acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k] *= 0.5f;
            store4(dPb + (size_t)i * a.ldp + col + c4, acc);
        }
        return;
    }
    task -= n_rel;
    const bool is_box = task == 0;
    const int col = is_box ? a.col_box : a.col_attr, off = is_box ? a.off_box : a.off_attr;
    float cc[4], gi[4] = {0.f, 0.f, 0.f, 0.f};
    load4(C + (size_t)b * a.ldp + col + c4, cc);
    if (is_box && a.off_img >= 0) {     // img = mean_r box: every box row receives g_img / R
        load4(gb + (size_t)a.off_img * H + c4, gi);
        const float inv = 1.f / (float)R;
#pragma unroll
        for (int k = 0; k < 4; ++k) gi[k] *= inv;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = jl; r < R; r += jstep) {
        float gv[4];
        load4(Pb + (size_t)r * a.ldp + col + c4, q);
        load4(gb + ((size_t)off + r) * H + c4, gv);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gv[k] = (q[k] + cc[k] > 0.f ? 1.f : a.slope) * (gv[k] + gi[k]);
            acc[k] += stored<A>(gv[k]);
        }
        store4(dPb + (size_t)r * a.ldp + col + c4, gv);
    }
    if (a.dC != nullptr) {   // no relation factor: this workgroup holds all R rows of its column block -> the per-image sum dC[b] here (fixed order)
        red[threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        __syncthreads();
        if (jl == 0) {
            for (int k = 1; k < jstep; ++k) {
                const float4 t = red[threadIdx.x + k * tpr];
                acc[0] += t.x; acc[1] += t.y; acc[2] += t.z; acc[3] += t.w;
            }
            store4((A*)a.dC + (size_t)b * a.ldp + col + c4, acc);
        }
    }
}

// dC[b, c] = sum_r dP[b R + r, c] (fixed order), c < cols: one thread per (b, 4 columns)
template <typename A>
__global__ __launch_bounds__(kEncThreads) void vis_segsum_kernel(const A* __restrict__ dP, A* __restrict__ dC, int B, int R, int cols, int ldp) {
    const int cv = cols >> 2;
    const size_t i = (size_t)blockIdx.x * kEncThreads + threadIdx.x;
    if (i >= (size_t)B * cv) return;
    const int b = (int)(i / cv), c = (int)(i - (size_t)b * cv) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, q[4];
    for (int r = 0; r < R; ++r) {
        load4(dP + ((size_t)b * R + r) * ldp + c, q);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += q[k];
    }
    store4(dC + (size_t)b * ldp + c, acc);
}

int enc_dtype_ok(const char* what, int dtype) {
    if (dtype != VLG_F32 && dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "%s: dtype %d", what, dtype);
    return 0;
}

}  // namespace

}  // namespace vlg

extern "C" {

int vlg_dropout(const void* x, const float* mask, int shared_rows, const uint64_t* rng, unsigned site, float p, const void* add, void* out, long long rows,
                int cols, int dtype, int out_dtype, void* stream) {
    using namespace vlg;
    if (int rc = enc_dtype_ok("dropout", dtype)) return rc;
    if (int rc = enc_dtype_ok("dropout", out_dtype)) return rc;
    if (rows < 0 || cols < 8 || cols % 8) return set_error(VLG_ERR_SHAPE, "dropout: rows=%lld cols=%d (cols must be a positive multiple of 8)", rows, cols);
    if (rows == 0) return 0;
    if (!x || !out) return set_error(VLG_ERR_ARG, "dropout: null buffer");
    if ((mask == nullptr) == (rng == nullptr)) return set_error(VLG_ERR_ARG, "dropout: exactly one of mask / rng");
    if (shared_rows < 0 || (shared_rows && !mask)) return set_error(VLG_ERR_ARG, "dropout: shared_rows=%d needs an explicit mask", shared_rows);
    if (!(p >= 0.f && p < 1.f)) return set_error(VLG_ERR_ARG, "dropout: p=%f outside [0, 1)", (double)p);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(add)) & 15)
        return set_error(VLG_ERR_ARG, "dropout: buffers must be 16-byte aligned");
    const uint32_t thr = drop_threshold(p);                       // keep <=> 16 random bits >= thr: p in steps of 2^-16
    const float scale = drop_scale(p);
    const size_t n = (size_t)rows * (cols >> 3);
    const dim3 grid((unsigned)((n + kEncThreads - 1) / kEncThreads));
    hipStream_t s = (hipStream_t)stream;
#define VLG_DROP(A, O) hipLaunchKernelGGL((dropout_kernel<A, O>), grid, dim3(kEncThreads), 0, s, (const A*)x, mask, shared_rows, rng, site, thr, scale, (const O*)add, (O*)out, (size_t)rows, cols)
    if (dtype == VLG_F32 && out_dtype == VLG_F32) VLG_DROP(float, float);
    else if (dtype == VLG_F32) VLG_DROP(float, uint16_t);
    else if (out_dtype == VLG_F32) VLG_DROP(uint16_t, float);
    else VLG_DROP(uint16_t, uint16_t);
#undef VLG_DROP
    return check_launch("dropout_kernel");
}

int vlg_dropout_mask(const uint64_t* rng, unsigned site, float p, float* out, long long n, void* stream) {
    using namespace vlg;
    if (n < 0 || !(p >= 0.f && p < 1.f)) return set_error(VLG_ERR_ARG, "dropout_mask: n=%lld p=%f", n, (double)p);
    if (n == 0) return 0;
    if (!rng || !out) return set_error(VLG_ERR_ARG, "dropout_mask: null buffer");
    const size_t groups = ((size_t)n + 7) / 8;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((groups + kEncThreads - 1) / kEncThreads)), dim3(kEncThreads), 0, (hipStream_t)stream, rng, site,
                       drop_threshold(p), drop_scale(p), out, (size_t)n);
    return check_launch("dropout_mask_kernel");
}

int vlg_rng_advance(uint64_t* rng, void* stream) {
    if (!rng) return vlg::set_error(VLG_ERR_ARG, "rng_advance: null state");
    hipLaunchKernelGGL(vlg::rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng);
    return vlg::check_launch("rng_advance_kernel");
}

static int vis_check(const char* what, int B, int R, int H, int V, int ldp, int dtype, int col_box, int col_rel, int col_attr, int off_box, int off_rel,
                     int off_attr, int off_img) {
    using namespace vlg;
    if (int rc = enc_dtype_ok(what, dtype)) return rc;
    if (B < 1 || R < 1 || H < 4 || H % 4 || H > 1024 || kEncThreads % (H / 4)) return set_error(VLG_ERR_SHAPE, "%s: B=%d R=%d H=%d (H/4 must divide 256)", what, B, R, H);
    if (B > 65535) return set_error(VLG_ERR_SHAPE, "%s: B=%d exceeds grid.y", what, B);
    if (ldp % 8 || col_box < 0 || off_box < 0) return set_error(VLG_ERR_SHAPE, "%s: ldp=%d col_box=%d off_box=%d", what, ldp, col_box, off_box);
    if ((off_rel >= 0) != (col_rel >= 0) || (off_attr >= 0) != (col_attr >= 0)) return set_error(VLG_ERR_ARG, "%s: a factor needs both its column and its row offset", what);
    int need = off_box + R;
    for (int c : {col_box, col_rel, col_attr})
        if (c >= 0 && (c % 8 || c + H > ldp)) return set_error(VLG_ERR_SHAPE, "%s: column block %d..%d outside ldp=%d / not a multiple of 8", what, c, c + H, ldp);
    if (off_rel >= 0) need = need > off_rel + R * R ? need : off_rel + R * R;
    if (off_attr >= 0) need = need > off_attr + R ? need : off_attr + R;
    if (off_img >= 0) need = need > off_img + 1 ? need : off_img + 1;
    if (need > V) return set_error(VLG_ERR_SHAPE, "%s: the factors need %d rows, V=%d", what, need, V);
    return 0;
}

int vlg_vis_encoder(const void* P, const void* C, int B, int R, int H, int V, int ldp, int col_box, int col_rel, int col_attr, int off_box, int off_rel,
                    int off_attr, int off_img, int dtype, float slope, void* mid, void* stream) {
    using namespace vlg;
    if (int rc = vis_check("vis_encoder", B, R, H, V, ldp, dtype, col_box, col_rel, col_attr, off_box, off_rel, off_attr, off_img)) return rc;
    if (!P || !C || !mid) return set_error(VLG_ERR_ARG, "vis_encoder: null buffer");
    const VisArgs a{P, C, mid, R, H, V, ldp, col_box, col_rel, col_attr, off_box, off_rel, off_attr, off_img, slope};
    const dim3 grid((off_rel >= 0 ? R : 0) + 1 + (off_attr >= 0 ? 1 : 0), B);
    if (dtype == VLG_F32) hipLaunchKernelGGL(vis_encoder_fwd_kernel<float>, grid, dim3(kEncThreads), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(vis_encoder_fwd_kernel<uint16_t>, grid, dim3(kEncThreads), 0, (hipStream_t)stream, a);
    return check_launch("vis_encoder_fwd_kernel");
}

int vlg_vis_encoder_backward(const void* P, const void* C, const void* grad_mid, int B, int R, int H, int V, int ldp, int col_box, int col_rel, int col_attr,
                             int off_box, int off_rel, int off_attr, int off_img, int dtype, float slope, void* dP, void* dC, void* stream) {
    using namespace vlg;
    if (int rc = vis_check("vis_encoder_backward", B, R, H, V, ldp, dtype, col_box, col_rel, col_attr, off_box, off_rel, off_attr, off_img)) return rc;
    if (!P || !C || !grad_mid || !dP || !dC) return set_error(VLG_ERR_ARG, "vis_encoder_backward: null buffer");
    const VisBwdArgs a{P, C, grad_mid, dP, off_rel >= 0 ? nullptr : dC, R, H, V, ldp, col_box, col_rel, col_attr, off_box, off_rel, off_attr, off_img, slope};
    const dim3 grid((off_rel >= 0 ? R : 0) + 1 + (off_attr >= 0 ? 1 : 0), B);
    hipStream_t s = (hipStream_t)stream;
    if (off_rel < 0) {    // one launch: the per-image sums come out of the workgroups that hold whole images
        if (dtype == VLG_F32) hipLaunchKernelGGL(vis_encoder_bwd_kernel<float>, grid, dim3(kEncThreads), 0, s, a);
        else hipLaunchKernelGGL(vis_encoder_bwd_kernel<uint16_t>, grid, dim3(kEncThreads), 0, s, a);
        return check_launch("vis_encoder_bwd_kernel");
    }
    // the per-image sums run over the columns [0, cols): the encoders' blocks must be adjacent from column 0 in the order box | rel | attr
    // (what the Python mirror lays out; a C caller with another layout must hear about it -- ADVICE r05)
    if (col_box != 0 || col_rel != H || (off_attr >= 0 && col_attr != 2 * H))
        return set_error(VLG_ERR_SHAPE, "vis_encoder_backward: with a relation factor the column blocks must be box | rel | attr from column 0 "
                                        "(col_box=%d col_rel=%d col_attr=%d, H=%d)", col_box, col_rel, col_attr, H);
    const int cols = H * (1 + (off_rel >= 0) + (off_attr >= 0));
    const size_t n = (size_t)B * (cols >> 2);
    if (dtype == VLG_F32) {
        hipLaunchKernelGGL(vis_encoder_bwd_kernel<float>, grid, dim3(kEncThreads), 0, s, a);
        if (int rc = check_launch("vis_encoder_bwd_kernel")) return rc;
        hipLaunchKernelGGL(vis_segsum_kernel<float>, dim3((unsigned)((n + kEncThreads - 1) / kEncThreads)), dim3(kEncThreads), 0, s, (const float*)dP, (float*)dC, B, R, cols, ldp);
    } else {
        hipLaunchKernelGGL(vis_encoder_bwd_kernel<uint16_t>, grid, dim3(kEncThreads), 0, s, a);
        if (int rc = check_launch("vis_encoder_bwd_kernel")) return rc;
        hipLaunchKernelGGL(vis_segsum_kernel<uint16_t>, dim3((unsigned)((n + kEncThreads - 1) / kEncThreads)), dim3(kEncThreads), 0, s, (const uint16_t*)dP, (uint16_t*)dC, B, R, cols,
                           ldp);
    }
    return check_launch("vis_segsum_kernel");
}

}  // extern "C"
