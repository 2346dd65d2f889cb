// vlg_decode.hip -- the tensor half of decode_grounding_on_factor (src/model/joint.py:512-596) on the fused alignment
// outputs (gfx950).
//
//   The reference takes the batch diagonal of the [B,A,Q,V] alignment tensor, edits that copy in place (POS prior
//   :528-552, box heuristics :554-594) and sorts every query row to keep its five best columns (:596); separately it
//   takes max over V, then arg-max over images (:520).  Here the diagonal block and the max over V come out of
//   vlg_bilinear_align directly (no 774 MB tensor), and this kernel does the rest:
//     phase 0  factor2img[b,q] = first arg-max over a of maxV[b,a,q]
//     phase 1  x -= pen[b,q,seg(v)];  per row: the row maximum and the first maximum among the box columns [0, n_box);
//              a row whose best column is a box (and is > -1e5) marks that box as selected for the sentence
//              (rows >= n_word_rows do not vote for the relation mask, joint.py:571)
//     phase 2  relation columns (i,j): -100 unless both boxes are selected, -1e10 on i == j; attribute columns: -1e10
//              unless the box is selected; the row is written back (the reference's in-place semantics) and its five
//              largest entries are extracted in descending order, equal values by ascending column.
//   One block per sentence, one wave per query row at a time; lane l owns columns l, l+64, ... of its row in every
//   phase, so no synchronisation is needed between writing a row back and re-reading it; the five selection rounds
//   read the row from a per-wave LDS copy (global memory when 16 rows of V floats exceed the LDS budget).  All arithmetic is the
//   reference's fp32 arithmetic in the same order per element: values are bit-identical, only the order of EQUAL
//   values in the top five is a choice (torch.argsort is not stable either).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vlg_common.h"

namespace vlg {

constexpr int kDecWaves = 16, kDecTop = 5, kDecChunk = 8;
constexpr size_t kDecRowLds = 128 * 1024;   // budget for the per-wave row buffers

struct DecodeArgs {
    float* logit;              // [B,Q,V]
    const float* pen;          // [B,Q,S] or null
    const uint8_t* seg_of_v;   // [V] or null
    int n_seg, B, Q, V;
    int use_heuristic, n_box, rel_off, attr_off, n_word_rows;
    const float* maxV;         // [B,A,Q] or null
    int A;
    int32_t* factor2img;       // [B,Q] or null
    int32_t* top;              // [B,Q,5]
    int row_in_lds;            // the waves keep their current row in LDS for the five selection rounds
    uint8_t* sel;              // [B][2][n_box] selection tables in global memory (two-launch form only)
};

// (value, column) ordering of the sort: larger value first, equal values by ascending column.
__device__ __forceinline__ bool before(float v, int i, float w, int j) { return v > w || (v == w && i < j); }

// MODE 0: one block per sentence does everything (tables in LDS).  Small batches leave most CUs idle that way, so the
// launcher can split a sentence's rows over gridDim.y blocks and the work over two launches around the one dependency --
// every row's vote must be in before any row is edited: MODE 1 = phases 0-1 (tables to global memory), MODE 2 = phase 2.
template <int MODE>
__global__ __launch_bounds__(64 * kDecWaves) void grounding_decode_kernel(DecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    uint8_t* sel_rel = reinterpret_cast<uint8_t*>(smem_raw);   // [n_box] box chosen by a word row (relation mask)
    uint8_t* sel_attr = sel_rel + max(p.n_box, 1);             // [n_box] box chosen by any row (attribute mask)
    float* rowbuf = reinterpret_cast<float*>(smem_raw + ((2 * (size_t)max(p.n_box, 1) + 15) & ~(size_t)15)) +
                    (size_t)(threadIdx.x >> 6) * p.V;          // [V] this wave's current row (if row_in_lds)
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Q = p.Q, V = p.V;
    const int q_first = blockIdx.y * kDecWaves + wave, q_step = kDecWaves * gridDim.y;
    uint8_t* gsel = MODE != 0 ? p.sel + (size_t)b * 2 * max(p.n_box, 1) : nullptr;
    float* xb = p.logit + (size_t)b * Q * V;
    const float* penb = p.pen ? p.pen + (size_t)b * Q * p.n_seg : nullptr;
    const bool heur = p.use_heuristic != 0 && p.n_box > 0;

    for (int i = threadIdx.x; i < 2 * max(p.n_box, 1); i += blockDim.x) sel_rel[i] = MODE == 2 ? gsel[i] : 0;
    if (MODE == 1) { sel_rel = gsel; sel_attr = gsel + max(p.n_box, 1); }   // votes go straight to the global tables (zeroed by the launcher)

    // phase 0: image of every query = first arg-max over images of max_v (joint.py:520).  P threads per query, each over
    // the images a = part, part + P, ...; the partial (value, image) pairs meet in LDS.
    if (MODE != 2 && blockIdx.y == 0 && p.factor2img && p.maxV) {
        __shared__ float part_v[64 * kDecWaves];
        __shared__ int part_a[64 * kDecWaves];
        const int P = Q <= (int)blockDim.x ? min((int)blockDim.x / Q, p.A) : 1;
        const float* base = p.maxV + (size_t)b * p.A * Q;
        for (int item = threadIdx.x; item < Q * P; item += blockDim.x) {
            const int part = item / Q, q = item - part * Q;
            float best = base[(size_t)part * Q + q];
            int at = part;
            for (int a = part + P; a < p.A; a += P) {
                const float v = base[(size_t)a * Q + q];
                if (v > best) { best = v; at = a; }
            }
            if (P == 1) p.factor2img[(size_t)b * Q + q] = at;
            else { part_v[item] = best; part_a[item] = at; }
        }
        if (P > 1) {
            __syncthreads();
            for (int q = threadIdx.x; q < Q; q += blockDim.x) {
                float best = part_v[q];
                int at = part_a[q];
                for (int k = 1; k < P; ++k)
                    if (before(part_v[k * Q + q], part_a[k * Q + q], best, at)) { best = part_v[k * Q + q]; at = part_a[k * Q + q]; }
                p.factor2img[(size_t)b * Q + q] = at;
            }
        }
    }
    __syncthreads();

    // phase 1: prior in place; row statistics; box selection
    for (int q = q_first; MODE != 2 && q < Q; q += q_step) {
        float* row = xb + (size_t)q * V;
        float rmax = -INFINITY, bmax = -INFINITY;
        int bat = 0x7fffffff;
        // kDecChunk columns per lane per trip, their reads issued together (one column at a time is a chain of three
        // dependent reads -- value, segment, prior -- per 64 columns)
        for (int v0 = lane; v0 < V; v0 += 64 * kDecChunk) {
            float x[kDecChunk], pn[kDecChunk];
#pragma unroll
            for (int k = 0; k < kDecChunk; ++k) {
                const int v = min(v0 + 64 * k, V - 1);
                x[k] = row[v];
                pn[k] = penb ? penb[(size_t)q * p.n_seg + p.seg_of_v[v]] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < kDecChunk; ++k) {
                const int v = v0 + 64 * k;
                if (v < V) {
                    if (penb) {
                        x[k] -= pn[k];        // joint.py:547-550 (one subtraction per other segment, summed in pen)
                        row[v] = x[k];
                    }
                    rmax = fmaxf(rmax, x[k]);
                    if (heur && v < p.n_box && before(x[k], v, bmax, bat)) { bmax = x[k]; bat = v; }
                }
            }
        }
        if (heur) {
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) {
                rmax = fmaxf(rmax, __shfl_xor(rmax, k, 64));
                const float ov = __shfl_xor(bmax, k, 64);
                const int oi = __shfl_xor(bat, k, 64);
                if (before(ov, oi, bmax, bat)) { bmax = ov; bat = oi; }
            }
            // joint.py:568-571 / :583-585: the row's best column is a box, and it is a live one
            if (lane == 0 && bmax == rmax && bmax > -1e5f && bat < p.n_box) {
                sel_attr[bat] = 1;
                if (q < p.n_word_rows) sel_rel[bat] = 1;
            }
        }
    }
    __syncthreads();

    // phase 2: heuristics in place, then the five best columns of the row
    for (int q = q_first; MODE != 1 && q < Q; q += q_step) {
        float* row = xb + (size_t)q * V;
        const bool edit = heur && (p.rel_off >= 0 || p.attr_off >= 0);
        if (edit || p.row_in_lds)
            for (int v0 = lane; v0 < V; v0 += 64 * kDecChunk) {
                float xs[kDecChunk];
#pragma unroll
                for (int k = 0; k < kDecChunk; ++k) xs[k] = row[min(v0 + 64 * k, V - 1)];
#pragma unroll
                for (int k = 0; k < kDecChunk; ++k) {
                    const int v = v0 + 64 * k;
                    if (v >= V) continue;
                    float x = xs[k];
                    bool touched = false;
                    if (edit && p.rel_off >= 0 && v >= p.rel_off && v < p.rel_off + p.n_box * p.n_box) {
                        const int i = (v - p.rel_off) / p.n_box, j = (v - p.rel_off) - i * p.n_box;
                        if (!(sel_rel[i] && sel_rel[j])) x -= 100.f;   // joint.py:580
                        if (i == j) x = -1e10f;                        // joint.py:582
                        touched = true;
                    }
                    if (edit && p.attr_off >= 0 && v >= p.attr_off && v < p.attr_off + p.n_box) {
                        if (!sel_attr[v - p.attr_off]) x = -1e10f;     // joint.py:594
                        touched = true;
                    }
                    if (touched) row[v] = x;
                    if (p.row_in_lds) rowbuf[v] = x;
                }
            }
        const float* src = p.row_in_lds ? rowbuf : row;   // lane l wrote exactly the entries it reads back
        // five rounds of (max value, smallest column) strictly after the previous winner in the sort order
        float pv = INFINITY;
        int pi = -1;
        for (int r = 0; r < kDecTop; ++r) {
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            for (int v = lane; v < V; v += 64) {
                const float x = src[v];
                if (before(pv, pi, x, v) && before(x, v, bv, bi)) { bv = x; bi = v; }
            }
#pragma unroll
            for (int k = 1; k < 64; k <<= 1) {
                const float ov = __shfl_xor(bv, k, 64);
                const int oi = __shfl_xor(bi, k, 64);
                if (before(ov, oi, bv, bi)) { bv = ov; bi = oi; }
            }
            if (lane == 0) p.top[((size_t)b * Q + q) * kDecTop + r] = bi == 0x7fffffff ? -1 : bi;   // -1: fewer than 5 columns
            pv = bv;
            pi = bi;
        }
    }
}

}  // namespace vlg

extern "C" size_t vlg_grounding_decode_workspace(int B, int n_box) {
    if (B < 1) return 0;
    return ((size_t)B * 2 * (size_t)(n_box > 0 ? n_box : 1) + 255) & ~(size_t)255;
}

extern "C" int vlg_grounding_decode(float* logit, const float* pen, const uint8_t* seg_of_v, int n_seg, int B, int Q, int V,
                                    int use_heuristic, int n_box, int rel_offset, int attr_offset, int n_word_rows,
                                    const float* maxV, int A, int32_t* factor2img, int32_t* top5, void* ws, size_t ws_bytes,
                                    void* stream) {
    using namespace vlg;
    if (B < 0 || Q <= 0 || V <= 0) return set_error(VLG_ERR_SHAPE, "grounding_decode: B=%d Q=%d V=%d", B, Q, V);
    if (!logit || !top5) return set_error(VLG_ERR_ARG, "grounding_decode: logit and top5 are required");
    if ((pen != nullptr) != (seg_of_v != nullptr) || (pen && n_seg <= 0))
        return set_error(VLG_ERR_ARG, "grounding_decode: pen, seg_of_v and n_seg > 0 go together");
    if ((factor2img != nullptr) != (maxV != nullptr) || (maxV && A <= 0))
        return set_error(VLG_ERR_ARG, "grounding_decode: maxV, A > 0 and factor2img go together");
    if (use_heuristic) {
        if (n_box <= 0 || n_box > V) return set_error(VLG_ERR_SHAPE, "grounding_decode: n_box=%d with V=%d", n_box, V);
        if (rel_offset >= 0 && (long)rel_offset + (long)n_box * n_box > V)
            return set_error(VLG_ERR_SHAPE, "grounding_decode: relation block %d + %d^2 exceeds V=%d", rel_offset, n_box, V);
        if (attr_offset >= 0 && attr_offset + n_box > V)
            return set_error(VLG_ERR_SHAPE, "grounding_decode: attribute block %d + %d exceeds V=%d", attr_offset, n_box, V);
        if (n_box > 16384) return set_error(VLG_ERR_SHAPE, "grounding_decode: n_box=%d exceeds the selection table", n_box);
    }
    if (B == 0) return 0;
    DecodeArgs a{logit, pen, seg_of_v, n_seg, B, Q, V, use_heuristic, use_heuristic ? n_box : 0, rel_offset, attr_offset,
                 n_word_rows, maxV, A, factor2img, top5, 0, nullptr};
    size_t lds = (2 * (size_t)(a.n_box > 0 ? a.n_box : 1) + 15) & ~(size_t)15;
    const size_t rows = sizeof(float) * kDecWaves * (size_t)V;
    if (lds + rows <= kDecRowLds) {
        a.row_in_lds = 1;
        lds += rows;
    }
    // rows of one sentence over G blocks when the batch alone cannot fill the chip (needs the table scratch)
    const int row_groups = (Q + kDecWaves - 1) / kDecWaves;
    int G = B >= 256 ? 1 : (256 + B - 1) / B;
    if (G > row_groups) G = row_groups;
    if (G > 1 && (!ws || ws_bytes < vlg_grounding_decode_workspace(B, a.n_box))) G = 1;
    hipStream_t s = (hipStream_t)stream;
    auto attr = [&](const void* k) -> int {
        if (lds <= 64 * 1024) return 0;
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return e == hipSuccess ? 0 : set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
    };
    if (G == 1) {
        if (int rc = attr(reinterpret_cast<const void*>(grounding_decode_kernel<0>))) return rc;
        hipLaunchKernelGGL(grounding_decode_kernel<0>, dim3(B), dim3(64 * kDecWaves), lds, s, a);
    } else {
        a.sel = static_cast<uint8_t*>(ws);
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)B * 2 * (size_t)(a.n_box > 0 ? a.n_box : 1), s);
        if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
        if (int rc = attr(reinterpret_cast<const void*>(grounding_decode_kernel<1>))) return rc;
        if (int rc = attr(reinterpret_cast<const void*>(grounding_decode_kernel<2>))) return rc;
        hipLaunchKernelGGL(grounding_decode_kernel<1>, dim3(B, G), dim3(64 * kDecWaves), lds, s, a);
        hipLaunchKernelGGL(grounding_decode_kernel<2>, dim3(B, G), dim3(64 * kDecWaves), lds, s, a);
    }
    return check_launch("grounding_decode_kernel");
}
