// vlg_gemm.hip -- the tall-skinny GEMMs of the encoder projections around the contraction:
// the weight / bias gradient of an nn.Linear over all B*N token rows,
//     dW[out,in] = dY^T X,   db[out] = sum_rows dY          (MLP.linear, src/model/nn/common.py:30,47-51 under loss.backward();
//                                                             word / child / parent encoders, src/model/joint.py:270-277)
// The contraction runs over the ~10^4 token rows and the result is a few hundred columns square: a library GEMM maps it to
// (out/128) x (in/64) tiles = 4 workgroups on a 256-CU chip (74-81 us per call in the training step).  Here the rows are
// split over the whole chip (split-K), every workgroup streams its row chunk once through LDS, and the partial tiles are
// added in a FIXED order by a second launch -- no atomics, bit-reproducible.
//
// Both operands are stored row-major with the contraction index as the SLOW dimension ([rows][cols]); the MFMA wants 8
// consecutive contraction positions per lane.  gfx950's ds_read_b64_tr_b16 does that transposition on the way out of LDS:
// a 16-lane group reads a 4-row x 16-column block of 16-bit elements and every lane receives one column of it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "vlg_common.h"
#include "vlg_mfma.h"

namespace vlg {

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int kGemmThreads = 256;   // 4 wavefronts, 2 x 2 over the output tile
// contraction rows per LDS stage and register sets in flight of the two bf16 tile shapes (tools/build_variant_gemm.sh overrides them for A/B timing)
#ifndef VLG_TN64_STAGE
#define VLG_TN64_STAGE 128
#endif
#ifndef VLG_TN64_SETS
#define VLG_TN64_SETS 2
#endif
#ifndef VLG_TN128_STAGE
#define VLG_TN128_STAGE 64
#endif
#ifndef VLG_TN128_SETS
#define VLG_TN128_SETS 1
#endif
#ifndef VLG_TN_WGS
#define VLG_TN_WGS 256      // workgroups a product is split into (tiles x row splits)
#endif
#ifndef VLG_SG_U
#define VLG_SG_U 4          // small products: contraction chunks whose loads are all issued before the first product
#endif
#ifndef VLG_TN_BIG_MIN
#define VLG_TN_BIG_MIN 8    // 128-tiles an output must have to take the 128-tile kernel
#endif

// Tile shapes.  TILE = output tile edge (rows of C = columns of A; columns of C = columns of B), KSTAGE = contraction rows per LDS stage.
//   <64, 128>   round 3: the encoder projections ([B N, 256] x [B N, 128..384]: few output tiles, deep split)
//   <128, 64>   round 5: outputs of >= 128 x 128 (the parser's 256 x 256 weights over 4 B L rows, the text / visual encoders' 256 x 800 /
//               256 x 2048): a wave owns 64 x 64 -- 16 transposed LDS reads feed 16 MFMAs per 32 contraction rows where the 64-tile's 8 reads
//               feed 4 (it was LDS-read-bound), and every operand column block is streamed by half as many workgroups.
// LDS row pitch in bytes: TILE bf16 + 32 bytes.  A transposed read of one 32-lane half covers 8 consecutive rows x 32 bytes;
// (2 TILE + 32) r mod 256 = 160 r (TILE 64) / 32 r (TILE 128) mod 256 for r = 0..7 puts them on 8 disjoint 8-bank groups (64 banks x 4 bytes).
template <int TILE, int KSTAGE>
struct TnCfg {
    static constexpr int kTile = TILE, kStage = KSTAGE, kPitch = TILE * 2 + 32;
    static constexpr int kQ = TILE / 2, kF = TILE / 32;               // a wave's quadrant edge; 16-wide fragments per quadrant edge
    static constexpr int kSeg = TILE / 8;                             // 16-byte segments per tile row
    static constexpr int kRowsPerPass = kGemmThreads / kSeg, kPasses = KSTAGE / kRowsPerPass;
};

// C_part[s] (TILE x TILE tile) = sum over the rows of split s of A[k, m0..]^T B[k, n0..]; optionally the column sums of A / of B.
// grid = tiles x ceil(S / 8) x 8 workgroups (see the block order below); A [K, lda], B [K, ldb] bf16; part [S][M][N] fp32, part_cs [S][M] fp32 (or null).
// SETS: register sets of one stage each, loaded SETS stages ahead of their use (the 128-tile holds one: with its 64 accumulator registers a
// second set spills at the 256 registers that two resident workgroups per CU allow); CS / CSB: carry the column sums of A / of B.
template <int TILE, int KSTAGE, int SETS, bool CS, bool CSB>
__device__ __forceinline__ void tn_body(const uint16_t* __restrict__ A, int lda, const uint16_t* __restrict__ B, int ldb, int K, int M, int N, int KC, int S,
                                        float* __restrict__ part, float* __restrict__ part_cs, float* __restrict__ part_csb, const int bid, char* sA,
                                        char* sB) {
    using C = TnCfg<TILE, KSTAGE>;
    constexpr int kTile = C::kTile, kStage = C::kStage, kPitch = C::kPitch, kQ = C::kQ, kF = C::kF, kP = C::kPasses, kRP = C::kRowsPerPass;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MT = (M + kTile - 1) / kTile;   // M, N: multiples of 8 (16-byte row segments); the last tile of either side may be partial
    // XCD-aware block order: consecutive workgroup ids go round-robin over the eight XCDs (one L2 each), so id % 8 picks the XCD.  All
    // tiles of one row split s get ids with the same id % 8: the split's rows of A and B are then fetched into ONE L2 and shared by its
    // MT x NT tiles there, instead of every column block being pulled into several different L2s (grid (tiles, S) order).
    const int tiles = MT * ((N + kTile - 1) / kTile);
    const int xcd = bid & 7, j = bid >> 3;
    const int s = (j / tiles) * 8 + xcd, tile = j % tiles;
    if (s >= S) return;                        // (the last group of eight splits may be partial; block-uniform)
    const int mt = tile % MT, nt = tile / MT;
    const int m0 = mt * kTile, n0 = nt * kTile;
    const int k_begin = s * KC, k_end = min(K, k_begin + KC);
    const int wm = wave & 1, wn = wave >> 1;   // this wave's quadrant

    // global -> register staging: a tile row is kSeg x 16 bytes; 256 threads cover kRP rows per pass, kP passes per stage.  The reads are
    // BUFFER loads over this split's rows of each operand (descriptor: base = row k_begin, extent = the split's rows): a row past the split's
    // end or a column past a partial tile's edge is out of range and comes back as zeros -- no branch around any load, so the staging is
    // straight-line code whose outstanding reads the compiler can COUNT (with a condition around every load each stage waited for vmcnt(0),
    // i.e. for the set fetched last as well: one stage in flight, not SETS)
    const int c16 = tid % C::kSeg, r0 = tid / C::kSeg;
    const bool col_a = m0 + c16 * 8 < M, col_b = n0 + c16 * 8 < N;
    const int rows = k_end - k_begin;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)k_begin * lda), 0, rows * lda * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)k_begin * ldb), 0, rows * ldb * 2, 0x00020000);
    constexpr int kOut = 0x40000000;                                   // (a byte offset past any extent: the host bounds rows x ld x 2 below 2^30)
    const int vo_a = col_a ? (r0 * lda + m0 + c16 * 8) * 2 : kOut, vo_b = col_b ? (r0 * ldb + n0 + c16 * 8) * 2 : kOut;
    // register sets of one stage each (kP x 16 bytes per operand and thread), loaded SETS stages ahead of their use
    u32x4 ra[SETS][kP], rb[SETS][kP];
    auto fetch = [&](int set, int ks) {   // rows k_begin + ks + r0 + kRP p
#ifdef VLG_TN_HOT          // tools/ ablation (results are wrong): every stage re-reads the split's first stage (cache-resident operands)
        ks = 0;
#endif
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            ra[set][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_a, vo_a, (ks + kRP * p) * lda * 2, 0);
            rb[set][p] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, vo_b, (ks + kRP * p) * ldb * 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);     // (the scheduler sank the reads below the stage's products: issued a stage late)
    };
    auto stash = [&](int set) {
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            *reinterpret_cast<u32x4*>(sA + (r0 + kRP * p) * kPitch + c16 * 16) = ra[set][p];
            *reinterpret_cast<u32x4*>(sB + (r0 + kRP * p) * kPitch + c16 * 16) = rb[set][p];
        }
    };

    // transposed-read addresses: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3 of its 4 x 16 block; group g
    // takes rows 4g..4g+3 (first read) and 16+4g.. (second read) of a 32-row step -- the same row assignment on both operands
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
    const int row_off = (4 * g + q) * kPitch + p4 * 8;
    const char* a_base = sA + row_off + (wm * kQ) * 2;
    const char* b_base = sB + row_off + (wn * kQ) * 2;

    f32x4 acc[kF][kF] = {};
    f32x4 cs[CS ? kF : 1] = {};
    f32x4 csb[CSB ? kF : 1] = {};
    // the column sums ride on the matrix cores (A^T 1 / 1^T B).  Which workgroups carry them is BLOCK-uniform (the first tile column / row), and the
    // whole pipeline below is instantiated per case: inside it the extra products are unconditional, so that every stage is one basic block
    // (a wave-level condition around them cut the stage into pieces the scheduler could not move fragment reads across)
    const bool blk_cs = CS && part_cs != nullptr && nt == 0, blk_csb = CSB && part_csb != nullptr && mt == 0;
    const bool want_cs = blk_cs && wn == 0, want_csb = blk_csb && wm == 0;   // (the waves that store them)
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (v8i16){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

    auto pipeline = [&](auto do_cs, auto do_csb) {
        constexpr bool kCs = decltype(do_cs)::value, kCsb = decltype(do_csb)::value;
        constexpr int KK = kStage / 32;
        // one stage: the fragments of contraction step kk + 1 are read while the products of step kk run (two fragment sets)
        auto stage_mma = [&]() {
            bf16x8 fa[2][kF], fb[2][kF];
            auto read = [&](int kk) {
#pragma unroll
                for (int t = 0; t < kF; ++t) {
                    fa[kk & 1][t] = tr_frag(tr_read(a_base + kk * 32 * kPitch + t * 32), tr_read(a_base + (kk * 32 + 16) * kPitch + t * 32));
                    fb[kk & 1][t] = tr_frag(tr_read(b_base + kk * 32 * kPitch + t * 32), tr_read(b_base + (kk * 32 + 16) * kPitch + t * 32));
                }
            };
            read(0);
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {
                if (kk + 1 < KK) read(kk + 1);
#pragma unroll
                for (int i = 0; i < kF; ++i)
#pragma unroll
                    for (int j = 0; j < kF; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
                if constexpr (kCs) {
#pragma unroll
                    for (int i = 0; i < kF; ++i) cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[kk & 1][i], ones, cs[i], 0, 0, 0);
                }
                if constexpr (kCsb) {   // ones^T B: every row of the result tile is the column sum
#pragma unroll
                    for (int j = 0; j < kF; ++j) csb[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fb[kk & 1][j], csb[j], 0, 0, 0);
                }
            }
        };
        fetch(0, 0);
        if constexpr (SETS == 2) {
            // two stages per trip and NO exit between them: the trip is one basic block (with a `break` after the first stage the reads of the
            // set fetched there were sunk into the block behind it -- below the stage's products, a stage late); an odd last stage follows the loop
            fetch(1, kStage);                  // (past the split's end: zeros)
            const int n_stages = (rows + kStage - 1) / kStage;
            int ks = 0;
            for (int trip = 0; trip < (n_stages >> 1); ++trip, ks += 2 * kStage) {
                __syncthreads();   // the previous stage's fragment reads are done
                stash(0);
                __syncthreads();
                fetch(0, ks + 2 * kStage);
                stage_mma();
                __syncthreads();
                stash(SETS - 1);
                __syncthreads();
                fetch(SETS - 1, ks + 3 * kStage);
                stage_mma();
            }
            if (n_stages & 1) {
                __syncthreads();
                stash(0);
                __syncthreads();
                stage_mma();
            }
        } else {
            for (int ks = 0; ks < rows; ks += kStage) {
                __syncthreads();
                stash(0);
                __syncthreads();
                fetch(0, ks + kStage);
                stage_mma();
            }
        }
    };
    if constexpr (CS && CSB) {
        if (blk_cs && blk_csb) pipeline(std::true_type{}, std::true_type{});
        else if (blk_cs) pipeline(std::true_type{}, std::false_type{});
        else if (blk_csb) pipeline(std::false_type{}, std::true_type{});
        else pipeline(std::false_type{}, std::false_type{});
    } else if constexpr (CS) {
        if (blk_cs) pipeline(std::true_type{}, std::false_type{});
        else pipeline(std::false_type{}, std::false_type{});
    } else if constexpr (CSB) {
        if (blk_csb) pipeline(std::false_type{}, std::true_type{});
        else pipeline(std::false_type{}, std::false_type{});
    } else {
        pipeline(std::false_type{}, std::false_type{});
    }

    // accumulator tile: lane l, register r <-> row 4 (l >> 4) + r, column l & 15
    float* out = part + (size_t)s * M * N;
#pragma unroll
    for (int i = 0; i < kF; ++i)
#pragma unroll
        for (int j = 0; j < kF; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * kQ + i * 16 + 4 * g + r, col = n0 + wn * kQ + j * 16 + (lane & 15);
#ifdef VLG_TN_NOSTORE      // tools/ ablation (results are wrong): no partial tiles
                if (row < M && col < N && acc[i][j][r] == 123.456f) out[(size_t)row * N + col] = acc[i][j][r];
#else
                if (row < M && col < N) out[(size_t)row * N + col] = acc[i][j][r];
#endif
            }
    if (CS && want_cs && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < (CS ? kF : 1); ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * kQ + i * 16 + 4 * g + r;
                if (row < M) part_cs[(size_t)s * M + row] = cs[i][r];
            }
    }
    if (CSB && want_csb && lane < 16) {   // row 0 of the tile: lanes 0..15, register 0
#pragma unroll
        for (int j = 0; j < (CSB ? kF : 1); ++j) {
            const int col = n0 + wn * kQ + j * 16 + lane;
            if (col < N) part_csb[(size_t)s * N + col] = csb[j][0];
        }
    }
}

template <int TILE, int KSTAGE, int SETS, bool CS, bool CSB>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_tn_kernel(const uint16_t* __restrict__ A, int lda,
                                                               const uint16_t* __restrict__ B, int ldb, int K, int M,
                                                               int N, int KC, int S, float* __restrict__ part,
                                                               float* __restrict__ part_cs, float* __restrict__ part_csb) {
    using C = TnCfg<TILE, KSTAGE>;
    __shared__ __attribute__((aligned(16))) char sA[C::kStage * C::kPitch];
    __shared__ __attribute__((aligned(16))) char sB[C::kStage * C::kPitch];
    tn_body<TILE, KSTAGE, SETS, CS, CSB>(A, lda, B, ldb, K, M, N, KC, S, part, part_cs, part_csb, (int)blockIdx.x, sA, sB);
}

// SEVERAL split-K products in one launch (round 6): the weight gradients of one backward pass are leaves -- nothing downstream reads them --
// so a Function defers them to its end and issues them as one grid per tile shape: a product is only 256 workgroups (one per CU), and
// launched alone each pays the chip's fill and drain; in one grid the next product's workgroups start as the previous one's finish.
// The descriptors travel by value; workgroup x finds its product by a scan of the block prefix (uniform: scalar code).  Each item's block
// count is a multiple of 8, so its local block index keeps the XCD assignment of the single launch.
constexpr int kTnGroupMax = 8;
struct TnItem {
    const uint16_t *A, *B;
    float *part, *part_cs, *part_csb;
    int lda, ldb, K, M, N, KC, S;
};
struct TnGroup {
    TnItem p[kTnGroupMax];
    int start[kTnGroupMax + 1];
    int count;
};

template <int TILE, int KSTAGE, int SETS, bool CS, bool CSB>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_tn_group_kernel(const TnGroup g) {
    using C = TnCfg<TILE, KSTAGE>;
    __shared__ __attribute__((aligned(16))) char sA[C::kStage * C::kPitch];
    __shared__ __attribute__((aligned(16))) char sB[C::kStage * C::kPitch];
    int i = 0;
    while (i + 1 < g.count && (int)blockIdx.x >= g.start[i + 1]) ++i;
    const TnItem& t = g.p[i];
    tn_body<TILE, KSTAGE, SETS, CS, CSB>(t.A, t.lda, t.B, t.ldb, t.K, t.M, t.N, t.KC, t.S, t.part, t.part_cs, t.part_csb, (int)blockIdx.x - g.start[i], sA, sB);
}

// ---- float32 operands on the bf16 matrix cores (round 5: the reference's `precision: 32`, config/trainer/train.yaml:20) -----------------
// x = hi + lo + e with hi = bf16(x), lo = bf16(x - hi) (both round-to-nearest: v_cvt_pk_bf16_f32), |e| <= 2^-17 |x|: the product
// a b = a_hi b_hi + a_hi b_lo + a_lo b_hi + O(2^-16 |a b|) is three bf16 MFMAs into the same fp32 accumulator -- against the library's
// fp32 GEMM (v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate, and 4 output tiles on 256 CUs for these shapes).  The split happens between
// the staging registers and LDS (four LDS images: hi / lo of both operands), so the operands are read once, as float32.  Same split-K
// plan, same block order and the same partial-tile layout as gemm_tn_kernel: the reductions are shared.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f32x2_t v = {a, b};
    const bf16x2_t h = __builtin_convertvector(v, bf16x2_t);
    const f32x2_t r = v - __builtin_convertvector(h, f32x2_t);
    hi = __builtin_bit_cast(uint32_t, h);
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2_t));
}
__device__ __forceinline__ void split8(const f32x4& u, const f32x4& v, uint4& hi, uint4& lo) {
    split_pair(u[0], u[1], hi.x, lo.x);
    split_pair(u[2], u[3], hi.y, lo.y);
    split_pair(v[0], v[1], hi.z, lo.z);
    split_pair(v[2], v[3], hi.w, lo.w);
}

template <int TILE, int KSTAGE, bool CS, bool CSB>
__global__ __launch_bounds__(kGemmThreads, 2) void gemm_tn3_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                                                                int K, int M, int N, int KC, int S, float* __restrict__ part,
                                                                float* __restrict__ part_cs, float* __restrict__ part_csb) {
    using C = TnCfg<TILE, KSTAGE>;
    constexpr int kTile = C::kTile, kStage = C::kStage, kPitch = C::kPitch, kQ = C::kQ, kF = C::kF, kP = C::kPasses, kRP = C::kRowsPerPass;
    __shared__ __attribute__((aligned(16))) char sAh[kStage * kPitch];
    __shared__ __attribute__((aligned(16))) char sAl[kStage * kPitch];
    __shared__ __attribute__((aligned(16))) char sBh[kStage * kPitch];
    __shared__ __attribute__((aligned(16))) char sBl[kStage * kPitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int MT = (M + kTile - 1) / kTile;
    const int tiles = MT * ((N + kTile - 1) / kTile);
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;       // (block order: see gemm_tn_kernel)
    const int s = (j / tiles) * 8 + xcd, tile = j % tiles;
    if (s >= S) return;
    const int mt = tile % MT, nt = tile / MT;
    const int m0 = mt * kTile, n0 = nt * kTile;
    const int k_begin = s * KC, k_end = min(K, k_begin + KC);
    const int wm = wave & 1, wn = wave >> 1;

    const int c16 = tid % C::kSeg, r0 = tid / C::kSeg;        // a staging slot = 8 consecutive columns of one row: 32 bytes of float32
    const bool col_a = m0 + c16 * 8 < M, col_b = n0 + c16 * 8 < N;
    const float* pa = A + (size_t)(k_begin + r0) * lda + m0 + c16 * 8;
    const float* pb = B + (size_t)(k_begin + r0) * ldb + n0 + c16 * 8;
    f32x4 ra[kP][2], rb[kP][2];
    auto fetch = [&](int ks) {
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            const bool ok = k_begin + ks + r0 + kRP * p < k_end;
            const f32x4* qa = reinterpret_cast<const f32x4*>(pa + (size_t)(ks + kRP * p) * lda);
            const f32x4* qb = reinterpret_cast<const f32x4*>(pb + (size_t)(ks + kRP * p) * ldb);
            ra[p][0] = ok && col_a ? qa[0] : z;
            ra[p][1] = ok && col_a ? qa[1] : z;
            rb[p][0] = ok && col_b ? qb[0] : z;
            rb[p][1] = ok && col_b ? qb[1] : z;
        }
    };
    auto stash = [&]() {
#pragma unroll
        for (int p = 0; p < kP; ++p) {
            uint4 h, l;
            const int o = (r0 + kRP * p) * kPitch + c16 * 16;
            split8(ra[p][0], ra[p][1], h, l);
            *reinterpret_cast<uint4*>(sAh + o) = h;
            *reinterpret_cast<uint4*>(sAl + o) = l;
            split8(rb[p][0], rb[p][1], h, l);
            *reinterpret_cast<uint4*>(sBh + o) = h;
            *reinterpret_cast<uint4*>(sBl + o) = l;
        }
    };

    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
    const int a_off = (4 * g + q) * kPitch + p4 * 8 + (wm * kQ) * 2, b_off = (4 * g + q) * kPitch + p4 * 8 + (wn * kQ) * 2;

    f32x4 acc[kF][kF] = {};
    f32x4 cs[CS ? kF : 1] = {};
    f32x4 csb[CSB ? kF : 1] = {};
    const bool want_cs = CS && part_cs != nullptr && nt == 0 && wn == 0;   // wave-uniform
    const bool want_csb = CSB && part_csb != nullptr && mt == 0 && wm == 0;
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (v8i16){0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});
    auto frag = [&](const char* base, int off, int kk, int t) {
        return tr_frag(tr_read(base + off + kk * 32 * kPitch + t * 32), tr_read(base + off + (kk * 32 + 16) * kPitch + t * 32));
    };
    auto stage_mma = [&]() {
#pragma unroll
        for (int kk = 0; kk < kStage / 32; ++kk) {
            bf16x8 fah[kF], fal[kF], fbh[kF], fbl[kF];
#pragma unroll
            for (int t = 0; t < kF; ++t) {
                fah[t] = frag(sAh, a_off, kk, t);
                fal[t] = frag(sAl, a_off, kk, t);
                fbh[t] = frag(sBh, b_off, kk, t);
                fbl[t] = frag(sBl, b_off, kk, t);
            }
#pragma unroll
            for (int i = 0; i < kF; ++i)
#pragma unroll
                for (int jj = 0; jj < kF; ++jj) {   // the two small terms first
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], fbh[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbl[jj], acc[i][jj], 0, 0, 0);
                    acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], fbh[jj], acc[i][jj], 0, 0, 0);
                }
            if constexpr (CS) {
                if (want_cs) {
#pragma unroll
                    for (int i = 0; i < kF; ++i) {
                        cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[i], ones, cs[i], 0, 0, 0);
                        cs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fah[i], ones, cs[i], 0, 0, 0);
                    }
                }
            }
            if constexpr (CSB) {
                if (want_csb) {
#pragma unroll
                    for (int jj = 0; jj < kF; ++jj) {
                        csb[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fbl[jj], csb[jj], 0, 0, 0);
                        csb[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fbh[jj], csb[jj], 0, 0, 0);
                    }
                }
            }
        }
    };

    const int rows = k_end - k_begin;
    fetch(0);
    for (int ks = 0; ks < rows; ks += kStage) {
        __syncthreads();   // the previous stage's fragment reads are done
        stash();
        __syncthreads();
        if (ks + kStage < rows) fetch(ks + kStage);
        stage_mma();
    }

    float* out = part + (size_t)s * M * N;
#pragma unroll
    for (int i = 0; i < kF; ++i)
#pragma unroll
        for (int jj = 0; jj < kF; ++jj)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * kQ + i * 16 + 4 * g + r, col = n0 + wn * kQ + jj * 16 + (lane & 15);
                if (row < M && col < N) out[(size_t)row * N + col] = acc[i][jj][r];
            }
    if (CS && want_cs && (lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < (CS ? kF : 1); ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * kQ + i * 16 + 4 * g + r;
                if (row < M) part_cs[(size_t)s * M + row] = cs[i][r];
            }
    }
    if (CSB && want_csb && lane < 16) {
#pragma unroll
        for (int jj = 0; jj < (CSB ? kF : 1); ++jj) {
            const int col = n0 + wn * kQ + jj * 16 + lane;
            if (col < N) part_csb[(size_t)s * N + col] = csb[jj][0];
        }
    }
}

// out[i] = sum_s part[s][i] in the order s = 0, 1, ... over the n tile elements followed by the n_cs column sums (the
// partial column sums sit behind the partial tiles, split-major).  One element per thread: the loads of one thread are
// independent, so eight are in flight per lane; the additions keep the order.
__device__ __forceinline__ void st_out(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_out(uint16_t* p, float v) {   // bf16, round to nearest even
    const uint32_t u = __float_as_uint(v);
    *p = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <typename O>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ part, int S, int n, int N, int ld_out, O* __restrict__ out,
                                                          const float* __restrict__ part_cs, int n_cs, O* __restrict__ out_cs,
                                                          const float* __restrict__ part_csb, int n_csb, O* __restrict__ out_csb) {
    int i = blockIdx.x * 256 + threadIdx.x;
    const float* src;
    O* dst;
    size_t pitch;
    if (i < n) {   // the weight gradient: rows ld_out elements apart (a column block of a wider gradient tensor is written in place)
        src = part + i; pitch = n;
        dst = ld_out == N ? out + i : out + (size_t)(i / N) * ld_out + (i % N);
    }
    else if (i - n < n_cs) { i -= n; src = part_cs + i; dst = out_cs + i; pitch = n_cs; }
    else if (i - n - n_cs < n_csb) { i -= n + n_cs; src = part_csb + i; dst = out_csb + i; pitch = n_csb; }
    else return;
    float t = 0.f;
    int s = 0;
    for (; s + 8 <= S; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * pitch];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; s < S; ++s) t += src[(size_t)s * pitch];
    st_out(dst, t);
}


// The reductions of SEVERAL split-K products in one launch (round 5): a Function that issues many weight gradients whose results it only
// needs at its end (the parser's feed-forwards: seven of them) defers their second launches and flushes them together.
constexpr int kRedGroupMax = 12;
struct RedArgs {
    const float *part, *part_cs, *part_csb;
    void *out, *out_cs, *out_csb;
    int S, n, N, ld_out, n_cs, n_csb, bf16, start;   // start: first workgroup of this problem
};
struct RedGroup {
    RedArgs p[kRedGroupMax];
    int count, blocks;
};

template <typename O>
__device__ __forceinline__ void reduce_one(const RedArgs& a, int i) {
    const float* src;
    O* dst;
    size_t pitch;
    if (i < a.n) {
        src = a.part + i; pitch = a.n;
        dst = a.ld_out == a.N ? (O*)a.out + i : (O*)a.out + (size_t)(i / a.N) * a.ld_out + (i % a.N);
    } else if (i - a.n < a.n_cs) { i -= a.n; src = a.part_cs + i; dst = (O*)a.out_cs + i; pitch = a.n_cs; }
    else if (i - a.n - a.n_cs < a.n_csb) { i -= a.n + a.n_cs; src = a.part_csb + i; dst = (O*)a.out_csb + i; pitch = a.n_csb; }
    else return;
    float t = 0.f;
    int s = 0;
    for (; s + 8 <= a.S; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(s + u) * pitch];
#pragma unroll
        for (int u = 0; u < 8; ++u) t += v[u];
    }
    for (; s < a.S; ++s) t += src[(size_t)s * pitch];
    st_out(dst, t);
}

__global__ __launch_bounds__(256) void gemm_reduce_group_kernel(RedGroup g) {
    int k = 0;
    while (k + 1 < g.count && (int)blockIdx.x >= g.p[k + 1].start) ++k;
    const RedArgs& a = g.p[k];
    const int i = (blockIdx.x - a.start) * 256 + threadIdx.x;
    if (a.bf16) reduce_one<uint16_t>(a, i);
    else reduce_one<float>(a, i);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// Small (batched) products in weight space: C[z] = alpha * A[z] B[z] + bias + u v^T (+ C[z]) with M, N, K of a few hundred.  The
// library maps an output of 256 x 256 to ONE workgroup (tiles of 256 x 256: 15-28 us for 33-80 MFLOP on a 256-CU chip); here every
// 32 x 32 tile of every batch entry is one wavefront: 256 workgroups for the four folded bottleneck weights of the parser's mid_ff
// (W1 W0, `nn/dmv_spec.py:52-54`; vlgae_amd/parser_ff.py).  Operands are addressed through element strides (a transposed or
// column-sliced weight is read where it lies); each lane gathers its MFMA fragment with 2- or 4-byte loads (the operands are a few
// hundred KB and L2-resident; edges and K tails are zero-filled), fp32 accumulation, bf16 (v_mfma_f32_16x16x32_bf16) or exact fp32
// (v_mfma_f32_16x16x4_f32) products.
struct SgArgs {
    const void *a, *b, *bias, *u, *v;
    void* c;
    long long sab, sam, sak, sbb, sbk, sbn, scb, ldc, sbias, su, sv;
    int M, N, K, accumulate, veca, vecb;
    float alpha;
};

__device__ __forceinline__ float sg_ld(const float* p, long long i) { return p[i]; }
__device__ __forceinline__ float sg_ld(const uint16_t* p, long long i) { return __uint_as_float((uint32_t)p[i] << 16); }
// One MFMA fragment of an operand: EPL consecutive contraction positions k, k + 1, ... of one row / column.  `p` points at position k
// (the address is always in bounds: the caller clamps the row and only takes the fast paths on full chunks); `st` is the element
// stride between contraction positions, `vec` what the host found the operand to allow when st == 1: EPL = one 16-byte load,
// 2 = 4-byte loads (bf16 rows an even number of elements apart), 1 = element loads.  A gathered 2-byte load touches up to 64 cache
// lines per wave instruction (16 rows x 4 k-groups); the 16-byte form is one line per lane and an eighth of the instructions.
__device__ __forceinline__ void sg_frag(f32x4& f, const float* p, long long st, int vec) {
    if (vec == 4) {
        f = *reinterpret_cast<const f32x4*>(p);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = p[e * st];
    }
}
__device__ __forceinline__ void sg_frag(bf16x8& f, const uint16_t* p, long long st, int vec) {
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    v8i16 w;
    if (vec == 8) {
        w = *reinterpret_cast<const v8i16*>(p);
    } else if (vec == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t two = *reinterpret_cast<const uint32_t*>(p + 2 * e);
            w[2 * e] = (short)(two & 0xffffu);
            w[2 * e + 1] = (short)(two >> 16);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = (short)p[e * st];
    }
    f = __builtin_bit_cast(bf16x8, w);
}
__device__ __forceinline__ void sg_zero(f32x4& f) { f = f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ void sg_zero(bf16x8& f) {
    typedef short v8i16 __attribute__((ext_vector_type(8)));
    f = __builtin_bit_cast(bf16x8, v8i16{0, 0, 0, 0, 0, 0, 0, 0});
}
__device__ __forceinline__ void sg_set(f32x4& f, int e, const float* p, long long i, bool ok) { f[e] = ok ? p[i] : 0.f; }
__device__ __forceinline__ void sg_set(bf16x8& f, int e, const uint16_t* p, long long i, bool ok) {
    const uint16_t bits = ok ? p[i] : (uint16_t)0;
    f[e] = __builtin_bit_cast(__bf16, bits);
}
__device__ __forceinline__ void sg_st(float* p, long long i, float v) { p[i] = v; }
__device__ __forceinline__ void sg_st(uint16_t* p, long long i, float v) {
    const uint32_t u = __float_as_uint(v);
    p[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <bool F32IN, typename Out>
__device__ __forceinline__ void small_gemm_tile(const SgArgs& p, int tile, int z) {
    using Cfg = MfmaCfg<F32IN>;
    using T = typename Cfg::T;
    using Frag = typename Cfg::Frag;
    constexpr int KW = Cfg::KW, EPL = Cfg::EPL;
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + 31) / 32;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const T* A = static_cast<const T*>(p.a) + (long long)z * p.sab;
    const T* B = static_cast<const T*>(p.b) + (long long)z * p.sbb;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kfull = p.K / KW * KW;
    // rows / columns beyond the edge read a clamped (valid) address and are zeroed afterwards
    const T* arow[2];
    const T* bcol[2];
    bool aok[2], bok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = tm * 32 + i * 16 + r, col = tn * 32 + i * 16 + r;
        aok[i] = row < p.M;
        bok[i] = col < p.N;
        arow[i] = A + (long long)min(row, p.M - 1) * p.sam + (long long)(EPL * g) * p.sak;
        bcol[i] = B + (long long)min(col, p.N - 1) * p.sbn + (long long)(EPL * g) * p.sbk;
    }
    // the epilogue's operands (bias, the rank-one term's vectors) are requested HERE, ahead of the contraction: read behind it they were one more
    // dependent round trip of a wavefront that has its SIMD to itself
    Out* C = static_cast<Out*>(p.c) + (long long)z * p.scb;
    const T* bias = p.bias ? static_cast<const T*>(p.bias) + (long long)z * p.sbias : nullptr;
    const T* u = p.u ? static_cast<const T*>(p.u) + (long long)z * p.su : nullptr;
    const T* v = p.v ? static_cast<const T*>(p.v) + (long long)z * p.sv : nullptr;
    float e_b[2], e_v[2], e_u[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = min(tn * 32 + j * 16 + r, p.N - 1);
        e_b[j] = bias ? sg_ld(bias, col) : 0.f;
        e_v[j] = v ? sg_ld(v, col) : 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) e_u[j][n] = u ? sg_ld(u, min(tm * 32 + j * 16 + 4 * g + n, p.M - 1)) : 0.f;     // (index j = the row tile here)
    }
    int k0 = 0;
    constexpr int U = VLG_SG_U;                     // chunks whose loads are ALL issued before the first product: a chunk per dependent
    for (; k0 + U * KW <= kfull; k0 += U * KW) {   // L2 / HBM round trip made a 256 x 256 x 256 product 9.5 us (8 round trips)
        Frag af[U][2], bf[U][2];
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                sg_frag(af[q][i], arow[i] + (long long)(k0 + q * KW) * p.sak, p.sak, p.veca);
                sg_frag(bf[q][i], bcol[i] + (long long)(k0 + q * KW) * p.sbk, p.sbk, p.vecb);
                if (!aok[i]) sg_zero(af[q][i]);
                if (!bok[i]) sg_zero(bf[q][i]);
            }
#pragma unroll
        for (int q = 0; q < U; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mma_chunk<F32IN>(af[q][i], bf[q][j], acc[i][j]);
    }
    for (; k0 < kfull; k0 += KW) {                 // remaining full chunks: no bounds on k
        Frag af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            sg_frag(af[i], arow[i] + (long long)k0 * p.sak, p.sak, p.veca);
            sg_frag(bf[i], bcol[i] + (long long)k0 * p.sbk, p.sbk, p.vecb);
            if (!aok[i]) sg_zero(af[i]);
            if (!bok[i]) sg_zero(bf[i]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma_chunk<F32IN>(af[i], bf[j], acc[i][j]);
    }
    if (kfull < p.K) {                              // the ragged tail: element loads, zero beyond K
        Frag af[2], bf[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const bool kin = kfull + EPL * g + e < p.K;
                sg_set(af[i], e, arow[i], (long long)(kfull + (kin ? e : -EPL * g)) * p.sak, aok[i] && kin);
                sg_set(bf[i], e, bcol[i], (long long)(kfull + (kin ? e : -EPL * g)) * p.sbk, bok[i] && kin);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mma_chunk<F32IN>(af[i], bf[j], acc[i][j]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tn * 32 + j * 16 + r;
            if (col >= p.N) continue;
            const float bj = e_b[j], vj = e_v[j];
#pragma unroll
            for (int n = 0; n < 4; ++n) {   // accumulator layout: lane (r, g), register n <-> row 4 g + n, column r
                const int row = tm * 32 + i * 16 + 4 * g + n;
                if (row >= p.M) continue;
                float val = p.alpha * acc[i][j][n] + bj;
                if (u) val = fmaf(e_u[i][n], vj, val);
                const long long at = (long long)row * p.ldc + col;
                if (p.accumulate) val += sizeof(Out) == 4 ? reinterpret_cast<const float*>(C)[at] : __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(C)[at] << 16);
                sg_st(C, at, val);
            }
        }
}

template <bool F32IN, typename Out>
__global__ __launch_bounds__(64) void small_gemm_kernel(SgArgs p) {
    small_gemm_tile<F32IN, Out>(p, blockIdx.x, blockIdx.y);
}

// Several INDEPENDENT small products in one launch (round 5): the weight-space products of one dependency level of the parser's
// feed-forwards (the four folded bottlenecks, the folded projections, the token / root / decision MLP rows, the context term: eight
// launches of 4.5-15 us each, ~7 us of it launch + tail) as one grid.  The descriptors travel by value in the kernel argument
// (12 x 176 bytes); workgroup x finds its problem by a scan of the tile prefix (uniform: scalar code).
constexpr int kSgGroupMax = 12;
struct SgGroup {
    SgArgs p[kSgGroupMax];
    int start[kSgGroupMax + 1];      // first workgroup of problem i; start[count] = grid size
    int tiles[kSgGroupMax];          // tiles per batch entry
    unsigned char f32in[kSgGroupMax], f32out[kSgGroupMax];
    int count;
};

__global__ __launch_bounds__(64) void small_gemm_group_kernel(SgGroup gp) {
    int i = 0;
    while (i + 1 < gp.count && (int)blockIdx.x >= gp.start[i + 1]) ++i;
    const int local = blockIdx.x - gp.start[i], z = local / gp.tiles[i], tile = local - z * gp.tiles[i];
    const SgArgs& p = gp.p[i];
    if (gp.f32in[i]) {
        if (gp.f32out[i]) small_gemm_tile<true, float>(p, tile, z);
        else small_gemm_tile<true, uint16_t>(p, tile, z);
    } else {
        if (gp.f32out[i]) small_gemm_tile<false, float>(p, tile, z);
        else small_gemm_tile<false, uint16_t>(p, tile, z);
    }
}

struct TnPlan {
    int KC, S;
    size_t bytes;
};

// the tile shape of a product: the 128-tile once both output dimensions fill one (VLG_WGRAD_TILE64 forces the round-3 kernel: A/B timing)
// (with both column sums wanted: see wgrad_launch).  Below ~8 tiles of 128 the split count that fills the chip makes the partial tiles -- S x M x N
// floats written and read back -- cost more than the 64-tile's extra operand traffic: [4 B L, 256]^T [4 B L, 256] measured 23.0 vs 21.8 us.
inline bool tn_big(int M, int N) { return M >= 128 && N >= 128 && ((M + 127) / 128) * ((N + 127) / 128) >= VLG_TN_BIG_MIN && !VLG_ENV("VLG_WGRAD_TILE64"); }

TnPlan plan_tn(int K, int M, int N, bool big, bool f32 = false) {   // (float32 operands: half the stage depth -- four LDS images instead of two)
    const int kTile = big ? 128 : 64, kStage = f32 ? (big ? 32 : 64) : (big ? VLG_TN128_STAGE : VLG_TN64_STAGE);
    const int tiles = ((M + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
    // ~1 workgroup per CU.  (Rounds 3-5 ran 512: with a condition around every staged load only one stage was in flight and the second resident
    // workgroup hid the latency.  With counted waits 256 workgroups stream as fast and the partial tiles -- S x M x N floats written, then read
    // by the reduction -- halve: the training step 1.808 -> 1.797 ms; 320 / 384 were slower than either.)
    int S = (VLG_TN_WGS + tiles - 1) / tiles;
    const int max_s = (K + 2 * kStage - 1) / (2 * kStage);          // at least two stages per split: both register sets in flight from the start
    S = S < 1 ? 1 : (S > max_s ? max_s : S);
    int KC = ((K + S - 1) / S + kStage - 1) / kStage * kStage;      // whole stages per split
    S = (K + KC - 1) / KC;
    return {KC, S, sizeof(float) * (size_t)S * ((size_t)M * N + M + N)};
}

}  // namespace

}  // namespace vlg

extern "C" {

size_t vlg_linear_wgrad_workspace(int K, int M, int N) {
    if (K < 1 || M < 8 || N < 8 || M % 8 || N % 8) return 0;
    size_t need = 0;                                                   // the largest of the plans any operand type / tile shape may take
    for (int f32 = 0; f32 < 2; ++f32)
        for (int big = 0; big < (vlg::tn_big(M, N) ? 2 : 1); ++big) {
            const size_t b = vlg::plan_tn(K, M, N, big != 0, f32 != 0).bytes;
            need = b > need ? b : need;
        }
    return need;
}

static int wgrad_launch(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, int in_dtype, void* ws, size_t ws_bytes, int out_dtype,
                        void* d_weight, int ld_dw, void* d_bias, void* x_colsum, bool reduce_now, void* stream);

// one bf16 product of a grouped split-K launch: wgrad_launch's checks and plan -> the kernel's descriptor and its image class
static int wgrad_item(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, void* ws, size_t ws_bytes, bool want_bias, bool want_colsum,
                      vlg::TnItem& t, int& cls) {
    using namespace vlg;
    if (K < 1 || M < 8 || N < 8 || M % 8 || N % 8)
        return set_error(VLG_ERR_SHAPE, "linear_wgrad: need K >= 1 and M, N positive multiples of 8 (got K=%d M=%d N=%d)", K, M, N);
    if (ld_dy < M || ld_x < N || ld_dy % 8 || ld_x % 8)
        return set_error(VLG_ERR_SHAPE, "linear_wgrad: row strides must cover the columns and be multiples of 8 elements (ld_dy=%d ld_x=%d)", ld_dy, ld_x);
    if (!dy || !x || !ws) return set_error(VLG_ERR_ARG, "linear_wgrad: null buffer");
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return set_error(VLG_ERR_ARG, "linear_wgrad: dy, x and the workspace must be 16-byte aligned");
    const bool big = tn_big(M, N) && !(want_bias && want_colsum);
    const TnPlan pl = plan_tn(K, M, N, big, false);
    if (ws_bytes < pl.bytes) return set_error(VLG_ERR_WORKSPACE, "linear_wgrad: needs a %zu-byte workspace (got %zu)", pl.bytes, ws_bytes);
    if ((long long)pl.KC * std::max(ld_dy, ld_x) * 2 >= (1LL << 30))
        return set_error(VLG_ERR_SHAPE, "linear_wgrad: a row split of %d rows x %d elements exceeds 1 GiB", pl.KC, std::max(ld_dy, ld_x));
    float* part = (float*)ws;
    t.A = (const uint16_t*)dy; t.B = (const uint16_t*)x; t.lda = ld_dy; t.ldb = ld_x; t.K = K; t.M = M; t.N = N; t.KC = pl.KC; t.S = pl.S;
    t.part = part;
    t.part_cs = want_bias ? part + (size_t)pl.S * M * N : nullptr;
    t.part_csb = want_colsum ? part + (size_t)pl.S * ((size_t)M * N + M) : nullptr;
    cls = !big ? 0 : (want_bias ? 1 : (want_colsum ? 2 : 3));
    return 0;
}

int vlg_linear_wgrad(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, int in_dtype, void* ws, size_t ws_bytes, int out_dtype,
                     void* d_weight, int ld_dw, void* d_bias, void* x_colsum, void* stream) {
    return wgrad_launch(dy, ld_dy, x, ld_x, K, M, N, in_dtype, ws, ws_bytes, out_dtype, d_weight, ld_dw, d_bias, x_colsum, true, stream);
}

int vlg_linear_wgrad_partial(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, int in_dtype, void* ws, size_t ws_bytes,
                             int want_bias, int want_x_colsum, void* stream) {
    // the split-K launch alone: the partial tiles stay in `ws` until vlg_linear_wgrad_reduce_group adds them (ws must stay alive and untouched)
    static char dummy;
    return wgrad_launch(dy, ld_dy, x, ld_x, K, M, N, in_dtype, ws, ws_bytes, VLG_F32, &dummy, N, want_bias ? &dummy : nullptr, want_x_colsum ? &dummy : nullptr, false,
                        stream);
}

int vlg_linear_wgrad_partial_group(const VlgWgradPartial* items, int count, void* stream) {
    using namespace vlg;
    if (count < 0 || (count && !items)) return set_error(VLG_ERR_ARG, "linear_wgrad_partial_group: count=%d", count);
    static char dummy;
    hipStream_t s = (hipStream_t)stream;
    TnGroup grp[4];                      // by kernel image: 64-tile (both column sums) | 128-tile with the bias sum | with x's column sum | with none
    for (auto& g : grp) { g.count = 0; g.start[0] = 0; }
    auto flush = [&](int c) -> int {
        TnGroup& g = grp[c];
        if (g.count == 0) return 0;
        const dim3 grid(g.start[g.count]);
        if (c == 0) hipLaunchKernelGGL((gemm_tn_group_kernel<64, VLG_TN64_STAGE, VLG_TN64_SETS, true, true>), grid, dim3(kGemmThreads), 0, s, g);
        else if (c == 1) hipLaunchKernelGGL((gemm_tn_group_kernel<128, VLG_TN128_STAGE, VLG_TN128_SETS, true, false>), grid, dim3(kGemmThreads), 0, s, g);
        else if (c == 2) hipLaunchKernelGGL((gemm_tn_group_kernel<128, VLG_TN128_STAGE, VLG_TN128_SETS, false, true>), grid, dim3(kGemmThreads), 0, s, g);
        else hipLaunchKernelGGL((gemm_tn_group_kernel<128, VLG_TN128_STAGE, VLG_TN128_SETS, false, false>), grid, dim3(kGemmThreads), 0, s, g);
        g.count = 0;
        return check_launch("gemm_tn_group_kernel");
    };
    for (int i = 0; i < count; ++i) {
        const VlgWgradPartial& q = items[i];
        void* bias = q.want_bias ? &dummy : nullptr;
        void* colsum = q.want_x_colsum ? &dummy : nullptr;
        if (q.in_dtype != VLG_BF16) {     // float32 operands: the three-product kernel has no grouped form -- its own launch
            if (int rc = wgrad_launch(q.dy, q.ld_dy, q.x, q.ld_x, q.K, q.M, q.N, q.in_dtype, q.ws, q.ws_bytes, VLG_F32, &dummy, q.N, bias, colsum, false, stream)) return rc;
            continue;
        }
        TnItem t;
        int cls;
        if (int rc = wgrad_item(q.dy, q.ld_dy, q.x, q.ld_x, q.K, q.M, q.N, q.ws, q.ws_bytes, bias != nullptr, colsum != nullptr, t, cls)) return rc;
        TnGroup& g = grp[cls];
        const int kTile = cls == 0 ? 64 : 128;
        const int tiles = ((q.M + kTile - 1) / kTile) * ((q.N + kTile - 1) / kTile);
        g.p[g.count] = t;
        g.start[g.count + 1] = g.start[g.count] + tiles * ((t.S + 7) / 8) * 8;
        if (++g.count == kTnGroupMax)
            if (int rc = flush(cls)) return rc;
    }
    for (int c = 0; c < 4; ++c)
        if (int rc = flush(c)) return rc;
    return 0;
}

int vlg_linear_wgrad_reduce_group(const VlgWgradReduce* items, int count, void* stream) {
    using namespace vlg;
    if (count < 0 || (count && !items)) return set_error(VLG_ERR_ARG, "linear_wgrad_reduce_group: count=%d", count);
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += kRedGroupMax) {
        RedGroup g;
        g.count = 0;
        g.blocks = 0;
        for (int i = base; i < count && i < base + kRedGroupMax; ++i) {
            const VlgWgradReduce& q = items[i];
            if (q.K < 1 || q.M < 8 || q.N < 8 || q.M % 8 || q.N % 8 || q.ld_dw < q.N) return set_error(VLG_ERR_SHAPE, "linear_wgrad_reduce_group: item %d K=%d M=%d N=%d ld_dw=%d", i, q.K, q.M, q.N, q.ld_dw);
            if (q.out_dtype != VLG_F32 && q.out_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "linear_wgrad_reduce_group: out_dtype %d", q.out_dtype);
            if (!q.ws || !q.d_weight) return set_error(VLG_ERR_ARG, "linear_wgrad_reduce_group: null buffer");
            if (q.in_dtype != VLG_F32 && q.in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "linear_wgrad_reduce_group: in_dtype %d", q.in_dtype);
            const TnPlan pl = plan_tn(q.K, q.M, q.N, tn_big(q.M, q.N) && !(q.d_bias && q.x_colsum), q.in_dtype == VLG_F32);
            const float* part = (const float*)q.ws;
            RedArgs& a = g.p[g.count++];
            a.part = part;
            a.part_cs = q.d_bias ? part + (size_t)pl.S * q.M * q.N : nullptr;
            a.part_csb = q.x_colsum ? part + (size_t)pl.S * ((size_t)q.M * q.N + q.M) : nullptr;
            a.out = q.d_weight; a.out_cs = q.d_bias; a.out_csb = q.x_colsum;
            a.S = pl.S; a.n = q.M * q.N; a.N = q.N; a.ld_out = q.ld_dw; a.n_cs = q.d_bias ? q.M : 0; a.n_csb = q.x_colsum ? q.N : 0;
            a.bf16 = q.out_dtype == VLG_BF16;
            a.start = g.blocks;
            g.blocks += (a.n + a.n_cs + a.n_csb + 255) / 256;
        }
        if (g.count == 0) continue;
        hipLaunchKernelGGL(gemm_reduce_group_kernel, dim3(g.blocks), dim3(256), 0, s, g);
        if (int rc = check_launch("gemm_reduce_group_kernel")) return rc;
    }
    return 0;
}

static int wgrad_launch(const void* dy, int ld_dy, const void* x, int ld_x, int K, int M, int N, int in_dtype, void* ws, size_t ws_bytes, int out_dtype,
                        void* d_weight, int ld_dw, void* d_bias, void* x_colsum, bool reduce_now, void* stream) {
    using namespace vlg;
    if (in_dtype != VLG_F32 && in_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "linear_wgrad: in_dtype %d", in_dtype);
    if (out_dtype != VLG_F32 && out_dtype != VLG_BF16) return set_error(VLG_ERR_DTYPE, "linear_wgrad: out_dtype %d", out_dtype);
    const bool f32 = in_dtype == VLG_F32;
    if (K < 1 || M < 8 || N < 8 || M % 8 || N % 8)
        return set_error(VLG_ERR_SHAPE, "linear_wgrad: need K >= 1 and M, N positive multiples of 8 (got K=%d M=%d N=%d)", K, M, N);
    if (ld_dy < M || ld_x < N || ld_dy % 8 || ld_x % 8)
        return set_error(VLG_ERR_SHAPE, "linear_wgrad: row strides must cover the columns and be multiples of 8 elements (ld_dy=%d ld_x=%d)", ld_dy, ld_x);
    if (ld_dw < N) return set_error(VLG_ERR_SHAPE, "linear_wgrad: ld_dw=%d below N=%d", ld_dw, N);
    if (!dy || !x || !d_weight || !ws) return set_error(VLG_ERR_ARG, "linear_wgrad: null buffer");
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ws)) & 15)
        return set_error(VLG_ERR_ARG, "linear_wgrad: dy, x and the workspace must be 16-byte aligned");
    const bool big = tn_big(M, N) && !(d_bias && x_colsum);     // (the 128-tile carries one kind of column sum)
    const TnPlan pl = plan_tn(K, M, N, big, f32);
    if (ws_bytes < pl.bytes) return set_error(VLG_ERR_WORKSPACE, "linear_wgrad: needs a %zu-byte workspace (got %zu)", pl.bytes, ws_bytes);
    if (!f32 && (long long)pl.KC * std::max(ld_dy, ld_x) * 2 >= (1LL << 30))     // (the bf16 kernel's 32-bit buffer offsets over one split's rows)
        return set_error(VLG_ERR_SHAPE, "linear_wgrad: a row split of %d rows x %d elements exceeds 1 GiB", pl.KC, std::max(ld_dy, ld_x));
    hipStream_t s = (hipStream_t)stream;
    float* part = (float*)ws;
    float* part_cs = d_bias ? part + (size_t)pl.S * M * N : nullptr;
    float* part_csb = x_colsum ? part + (size_t)pl.S * ((size_t)M * N + M) : nullptr;
    const int kTile = big ? 128 : 64;
    const int tiles = ((M + kTile - 1) / kTile) * ((N + kTile - 1) / kTile);
    const dim3 grid(tiles * ((pl.S + 7) / 8) * 8);
#define VLG_TN(...) hipLaunchKernelGGL((gemm_tn_kernel<__VA_ARGS__>), grid, dim3(kGemmThreads), 0, s, (const uint16_t*)dy, ld_dy, (const uint16_t*)x, ld_x, K, M, N, \
                                       pl.KC, pl.S, part, part_cs, part_csb)
#define VLG_TN3(...) hipLaunchKernelGGL((gemm_tn3_kernel<__VA_ARGS__>), grid, dim3(kGemmThreads), 0, s, (const float*)dy, ld_dy, (const float*)x, ld_x, K, M, N, \
                                        pl.KC, pl.S, part, part_cs, part_csb)
    if (f32) {   // float32 operands: three bf16 products per pair (gemm_tn3_kernel)
        if (!big) VLG_TN3(64, 64, true, true);
        else if (d_bias) VLG_TN3(128, 32, true, false);
        else if (x_colsum) VLG_TN3(128, 32, false, true);
        else VLG_TN3(128, 32, false, false);
    } else if (!big) VLG_TN(64, VLG_TN64_STAGE, VLG_TN64_SETS, true, true);
    else if (d_bias) VLG_TN(128, VLG_TN128_STAGE, VLG_TN128_SETS, true, false);
    else if (x_colsum) VLG_TN(128, VLG_TN128_STAGE, VLG_TN128_SETS, false, true);
    else VLG_TN(128, VLG_TN128_STAGE, VLG_TN128_SETS, false, false);
#undef VLG_TN
#undef VLG_TN3
    if (int rc = check_launch("gemm_tn_kernel")) return rc;
    if (!reduce_now) return 0;
    const int n = M * N, n_cs = d_bias ? M : 0, n_csb = x_colsum ? N : 0;
    if (out_dtype == VLG_F32)
        hipLaunchKernelGGL(gemm_reduce_kernel<float>, dim3((n + n_cs + n_csb + 255) / 256), dim3(256), 0, s, part, pl.S, n, N, ld_dw, (float*)d_weight, part_cs,
                           n_cs, (float*)d_bias, part_csb, n_csb, (float*)x_colsum);
    else
        hipLaunchKernelGGL(gemm_reduce_kernel<uint16_t>, dim3((n + n_cs + n_csb + 255) / 256), dim3(256), 0, s, part, pl.S, n, N, ld_dw, (uint16_t*)d_weight,
                           part_cs, n_cs, (uint16_t*)d_bias, part_csb, n_csb, (uint16_t*)x_colsum);
    return check_launch("gemm_reduce_kernel");
}

static int sg_make(vlg::SgArgs& p, const void* a, long long sab, long long sam, long long sak, const void* b, long long sbb, long long sbk, long long sbn,
                   void* c, long long scb, long long ldc, const void* bias, long long sbias, const void* u, long long su, const void* v,
                   long long sv, int batch, int M, int N, int K, float alpha, int accumulate, int in_dtype, int out_dtype) {
    using namespace vlg;
    if (batch < 0 || M < 1 || N < 1 || K < 1 || batch > 65535)
        return set_error(VLG_ERR_SHAPE, "small_gemm: bad shape batch=%d M=%d N=%d K=%d", batch, M, N, K);
    if ((in_dtype != VLG_F32 && in_dtype != VLG_BF16) || (out_dtype != VLG_F32 && out_dtype != VLG_BF16))
        return set_error(VLG_ERR_DTYPE, "small_gemm: dtypes %d -> %d", in_dtype, out_dtype);
    if (ldc < N) return set_error(VLG_ERR_SHAPE, "small_gemm: ldc=%lld below N=%d", ldc, N);
    if ((u == nullptr) != (v == nullptr)) return set_error(VLG_ERR_ARG, "small_gemm: the rank-one term needs both u and v");
    if (batch && (!a || !b || !c)) return set_error(VLG_ERR_ARG, "small_gemm: null buffer");
    // widest load per fragment of each operand (see sg_frag): its contraction index must be the unit-stride one
    const int epl = in_dtype == VLG_F32 ? 4 : 8;
    auto width = [&](const void* ptr, long long s_contract, long long s_other, long long s_batch) -> int {
        if (s_contract != 1) return 1;
        const uintptr_t addr = reinterpret_cast<uintptr_t>(ptr);
        if (s_other % epl == 0 && s_batch % epl == 0 && addr % 16 == 0) return epl;
        if (in_dtype == VLG_BF16 && s_other % 2 == 0 && s_batch % 2 == 0 && addr % 4 == 0) return 2;
        return 1;
    };
    p = SgArgs{a, b, bias, u, v, c, sab, sam, sak, sbb, sbk, sbn, scb, ldc, sbias, su, sv, M, N, K, accumulate,
               width(a, sak, sam, sab), width(b, sbk, sbn, sbb), alpha};
    return 0;
}

int vlg_small_gemm_group(const VlgSmallGemm* problems, int count, void* stream) {
    using namespace vlg;
    if (count < 0 || (count && !problems)) return set_error(VLG_ERR_ARG, "small_gemm_group: count=%d", count);
    hipStream_t s = (hipStream_t)stream;
    for (int base = 0; base < count; base += kSgGroupMax) {       // more than 12 problems: several launches of up to 12
        SgGroup gp;
        gp.count = 0;
        int blocks = 0;
        for (int i = base; i < count && i < base + kSgGroupMax; ++i) {
            const VlgSmallGemm& q = problems[i];
            SgArgs p;
            if (int rc = sg_make(p, q.a, q.sab, q.sam, q.sak, q.b, q.sbb, q.sbk, q.sbn, q.c, q.scb, q.ldc, q.bias, q.sbias, q.u, q.su, q.v, q.sv, q.batch,
                                 q.M, q.N, q.K, q.alpha, q.accumulate, q.in_dtype, q.out_dtype))
                return rc;
            if (q.batch == 0) continue;
            const int k = gp.count++;
            gp.p[k] = p;
            gp.tiles[k] = ((q.M + 31) / 32) * ((q.N + 31) / 32);
            gp.start[k] = blocks;
            gp.f32in[k] = q.in_dtype == VLG_F32;
            gp.f32out[k] = q.out_dtype == VLG_F32;
            blocks += gp.tiles[k] * q.batch;
        }
        if (gp.count == 0) continue;
        gp.start[gp.count] = blocks;
        hipLaunchKernelGGL(small_gemm_group_kernel, dim3(blocks), dim3(64), 0, s, gp);
        if (int rc = check_launch("small_gemm_group_kernel")) return rc;
    }
    return 0;
}

int vlg_small_gemm(const void* a, long long sab, long long sam, long long sak, const void* b, long long sbb, long long sbk, long long sbn,
                   void* c, long long scb, long long ldc, const void* bias, long long sbias, const void* u, long long su, const void* v,
                   long long sv, int batch, int M, int N, int K, float alpha, int accumulate, int in_dtype, int out_dtype, void* stream) {
    using namespace vlg;
    SgArgs p;
    if (int rc = sg_make(p, a, sab, sam, sak, b, sbb, sbk, sbn, c, scb, ldc, bias, sbias, u, su, v, sv, batch, M, N, K, alpha, accumulate, in_dtype, out_dtype))
        return rc;
    if (batch == 0) return 0;
    const dim3 grid(((M + 31) / 32) * ((N + 31) / 32), batch);
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == VLG_F32 && out_dtype == VLG_F32) hipLaunchKernelGGL((small_gemm_kernel<true, float>), grid, dim3(64), 0, s, p);
    else if (in_dtype == VLG_F32) hipLaunchKernelGGL((small_gemm_kernel<true, uint16_t>), grid, dim3(64), 0, s, p);
    else if (out_dtype == VLG_F32) hipLaunchKernelGGL((small_gemm_kernel<false, float>), grid, dim3(64), 0, s, p);
    else hipLaunchKernelGGL((small_gemm_kernel<false, uint16_t>), grid, dim3(64), 0, s, p);
    return check_launch("small_gemm_kernel");
}

}  // extern "C"
