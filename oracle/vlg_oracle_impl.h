/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU restatement ("oracle") of the VLGAE structured-DP hot path.  Included twice by
 * vlg_oracle.c, once with REAL=float (mirrors the reference's fp32 arithmetic) and once
 * with REAL=double (the high-precision truth the fp32 tolerances are stated against).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Every function cites the reference lines it restates (paths relative to /root/reference).
 * The reference expresses the DP with as_strided windows over (N+1)x(N+1)x2 charts and gets
 * the outside pass from autograd; here the same recurrences are written with explicit
 * indices, and the outside pass is the hand-derived adjoint of the inside loop, replayed
 * in reverse (each chart cell is written exactly once, so the charts are the tape).
 *
 * Parity status: PINNED -- checked against outputs of the reference itself
 * (tests/golden/<case>.npz, produced by tests/golden/make_golden.py importing /root/reference)
 * and against the reference's brute-force tree enumerator (deptree.py:213-228).
 */

#ifndef REAL
#error "define REAL, SUF, EXP, LOG before including"
#endif

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* src/model/torch_struct/dmv.py:7-15 */
#ifndef VLG_ORACLE_CONSTS
#define VLG_ORACLE_CONSTS
enum { HASCHILD = 0, NOCHILD = 1, LEFT = 0, RIGHT = 1, GO = 0, STOP = 1 };
enum { SR_LOG = 0, SR_MAX = 1 };
#endif

/* semiring "sum" over t[0..n): torch.logsumexp (semirings.py:131-132) or torch.max (:199-200).
 * Writes the normalised weights d(out)/d(t_r) into wgt when wgt != NULL:
 *   Log: exp(t_r - out)  (LogsumexpBackward)      Max: 1 at the FIRST maximal index, else 0. */
static REAL FN(sr_sum)(const REAL *t, int n, int semiring, REAL *wgt) {
    REAL m = t[0];
    int am = 0;
    for (int r = 1; r < n; ++r)
        if (t[r] > m) { m = t[r]; am = r; }
    if (semiring == SR_MAX) {
        if (wgt) for (int r = 0; r < n; ++r) wgt[r] = (r == am) ? (REAL)1 : (REAL)0;
        return m;
    }
    REAL s = 0;
    for (int r = 0; r < n; ++r) s += EXP(t[r] - m);
    REAL out = LOG(s) + m;
    if (wgt) for (int r = 0; r < n; ++r) wgt[r] = EXP(t[r] - out);
    return out;
}

/* ------------------------------------------------------------------------------------------
 * DMV1o inside + outside for ONE sentence.
 *   dec    [N,2(dir),2(val),2(decision)]   attach [N(head),N(child),2(val)]   (root-merged)
 * Chart layout is the reference's (dmv.py:33-35): I, C of shape (N+1)x(N+1)x2 where
 *   CL[h][l][v] = C[h][l][v]   (l <= h, complete, head h reaching left to l)
 *   CR[h][r][v] = C[h][r+1][v] (r >= h)            -- "diagonal(1) for right" (dmv.py:32)
 *   IL[h][c][v] = I[h][c][v]   (c <  h, incomplete, arc h -> c)
 *   IR[h][c][v] = I[h][c+1][v] (c >  h)
 * S keeps the two v-independent reductions of dmv.py:50,54 (needed as the lse normaliser of
 * the adjoint; recovering it as I - attach would cancel catastrophically at the sentinel).
 * ------------------------------------------------------------------------------------------ */
#define C_(h, j, v) Cc[(((h) * (N + 1)) + (j)) * 2 + (v)]
#define I_(h, j, v) Ic[(((h) * (N + 1)) + (j)) * 2 + (v)]
#define gC_(h, j, v) gC[(((h) * (N + 1)) + (j)) * 2 + (v)]
#define gI_(h, j, v) gI[(((h) * (N + 1)) + (j)) * 2 + (v)]
#define CL(h, l, v) C_(h, l, v)
#define CR(h, r, v) C_(h, (r) + 1, v)
#define IL(h, c, v) I_(h, c, v)
#define IR(h, c, v) I_(h, (c) + 1, v)
#define gCL(h, l, v) gC_(h, l, v)
#define gCR(h, r, v) gC_(h, (r) + 1, v)
#define gIL(h, c, v) gI_(h, c, v)
#define gIR(h, c, v) gI_(h, (c) + 1, v)
#define DEC(h, d, v, z) dec[(((h) * 2 + (d)) * 2 + (v)) * 2 + (z)]
#define ATT(h, c, v) attach[(((h) * N) + (c)) * 2 + (v)]
#define GDEC(h, d, v, z) gdec[(((h) * 2 + (d)) * 2 + (v)) * 2 + (z)]
#define GATT(h, c, v) gatt[(((h) * N) + (c)) * 2 + (v)]

static void FN(dmv1o_one)(const REAL *dec, const REAL *attach, int len, int N, int semiring, REAL neg_inf,
                          REAL glogZ, REAL *logZ, REAL *gdec, REAL *gatt, REAL *ws) {
    const int M = (N + 1) * (N + 1) * 2;
    REAL *Cc = ws, *Ic = ws + M, *gC = ws + 2 * M, *gI = ws + 3 * M;
    REAL *SLs = ws + 4 * M;               /* [N][N]: SL(i,j) at [j][i], SR(i,j) at [i][j] */
    REAL *t = SLs + N * N, *wg = t + N;   /* scratch, N each */

    for (int x = 0; x < M; ++x) { Cc[x] = neg_inf; Ic[x] = neg_inf; }       /* dmv.py:34-35 */
    for (int i = 0; i < N; ++i)
        for (int v = 0; v < 2; ++v) {                                        /* dmv.py:39-40 */
            CL(i, i, v) = DEC(i, LEFT, v, STOP);
            CR(i, i, v) = DEC(i, RIGHT, v, STOP);
        }

    for (int w = 1; w < N; ++w) {                                            /* dmv.py:47 */
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            /* dmv.py:50-52  IL[j][i][v] = (attach+dec[LEFT,GO])[j,i,v] + (+)_r CR[i][i+r][NOCHILD] + CL[j][i+r+1][HASCHILD] */
            for (int r = 0; r < w; ++r) t[r] = CR(i, i + r, NOCHILD) + CL(j, i + r + 1, HASCHILD);
            REAL sl = FN(sr_sum)(t, w, semiring, 0);
            SLs[j * N + i] = sl;
            for (int v = 0; v < 2; ++v) IL(j, i, v) = sl + (ATT(j, i, v) + DEC(j, LEFT, v, GO));
            /* dmv.py:54-56  IR[i][j][v] */
            for (int r = 0; r < w; ++r) t[r] = CR(i, i + r, HASCHILD) + CL(j, i + r + 1, NOCHILD);
            REAL sr = FN(sr_sum)(t, w, semiring, 0);
            SLs[i * N + j] = sr;
            for (int v = 0; v < 2; ++v) IR(i, j, v) = sr + (ATT(i, j, v) + DEC(i, RIGHT, v, GO));
        }
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            for (int v = 0; v < 2; ++v) {
                /* dmv.py:58-59  CL[j][i][v] = (+)_r CL[i+r][i][NOCHILD] + IL[j][i+r][v] */
                for (int r = 0; r < w; ++r) t[r] = CL(i + r, i, NOCHILD) + IL(j, i + r, v);
                CL(j, i, v) = FN(sr_sum)(t, w, semiring, 0);
                /* dmv.py:61-62  CR[i][j][v] = (+)_r IR[i][i+1+r][v] + CR[i+1+r][j][NOCHILD] */
                for (int r = 0; r < w; ++r) t[r] = IR(i, i + 1 + r, v) + CR(i + 1 + r, j, NOCHILD);
                CR(i, j, v) = FN(sr_sum)(t, w, semiring, 0);
            }
        }
        if (len != w) { CR(0, w, 0) = neg_inf; CR(0, w, 1) = neg_inf; }      /* dmv.py:63 single root */
    }
    *logZ = CR(0, len, NOCHILD);                                             /* dmv.py:65 */
    if (!gdec) return;

    /* ---- outside pass: adjoint of the loop above, replayed in reverse (reference: autograd) ---- */
    for (int x = 0; x < M; ++x) { gC[x] = 0; gI[x] = 0; }
    gCR(0, len, NOCHILD) = glogZ;
    for (int w = N - 1; w >= 1; --w) {
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            for (int v = 0; v < 2; ++v) {
                REAL g = gCL(j, i, v);
                if (g != 0) {
                    for (int r = 0; r < w; ++r) t[r] = CL(i + r, i, NOCHILD) + IL(j, i + r, v);
                    FN(sr_sum)(t, w, semiring, wg);
                    for (int r = 0; r < w; ++r) { gCL(i + r, i, NOCHILD) += g * wg[r]; gIL(j, i + r, v) += g * wg[r]; }
                }
                /* the masked cell was overwritten (index_put, dmv.py:63): no gradient reaches its inputs */
                g = (i == 0 && len != w) ? (REAL)0 : gCR(i, j, v);
                if (g != 0) {
                    for (int r = 0; r < w; ++r) t[r] = IR(i, i + 1 + r, v) + CR(i + 1 + r, j, NOCHILD);
                    FN(sr_sum)(t, w, semiring, wg);
                    for (int r = 0; r < w; ++r) { gIR(i, i + 1 + r, v) += g * wg[r]; gCR(i + 1 + r, j, NOCHILD) += g * wg[r]; }
                }
            }
        }
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            REAL gs = 0;
            for (int v = 0; v < 2; ++v) {
                REAL g = gIL(j, i, v);
                GATT(j, i, v) += g; GDEC(j, LEFT, v, GO) += g; gs += g;
            }
            if (gs != 0) {
                for (int r = 0; r < w; ++r) t[r] = CR(i, i + r, NOCHILD) + CL(j, i + r + 1, HASCHILD);
                FN(sr_sum)(t, w, semiring, wg);
                for (int r = 0; r < w; ++r) { gCR(i, i + r, NOCHILD) += gs * wg[r]; gCL(j, i + r + 1, HASCHILD) += gs * wg[r]; }
            }
            gs = 0;
            for (int v = 0; v < 2; ++v) {
                REAL g = gIR(i, j, v);
                GATT(i, j, v) += g; GDEC(i, RIGHT, v, GO) += g; gs += g;
            }
            if (gs != 0) {
                for (int r = 0; r < w; ++r) t[r] = CR(i, i + r, HASCHILD) + CL(j, i + r + 1, NOCHILD);
                FN(sr_sum)(t, w, semiring, wg);
                for (int r = 0; r < w; ++r) { gCR(i, i + r, HASCHILD) += gs * wg[r]; gCL(j, i + r + 1, NOCHILD) += gs * wg[r]; }
            }
        }
    }
    for (int i = 0; i < N; ++i)
        for (int v = 0; v < 2; ++v) {
            GDEC(i, LEFT, v, STOP) += gCL(i, i, v);
            GDEC(i, RIGHT, v, STOP) += gCR(i, i, v);
        }
}

/* Batched driver.  logZ [B]; gdec [B,N,2,2,2] / gatt [B,N,N,2] may be NULL (inside only);
 * glogZ [B] may be NULL (= ones).  Returns 0, or -1 on allocation failure / bad args.
 * Mirrors _Struct.sum + torch.autograd.grad(v.sum(), [dec, attach]) (helpers.py:101-116,150-154). */
int FN(orc_dmv1o)(const REAL *dec, const REAL *attach, const long long *lengths, int B, int N, int semiring,
                  double neg_inf, const REAL *glogZ, REAL *logZ, REAL *gdec, REAL *gatt) {
    if (B < 0 || N < 2) return -1;
    const size_t per = (size_t)4 * (N + 1) * (N + 1) * 2 + (size_t)N * N + 2 * (size_t)N;
    int fail = 0;
    if (gdec) memset(gdec, 0, sizeof(REAL) * (size_t)B * N * 8);
    if (gatt) memset(gatt, 0, sizeof(REAL) * (size_t)B * N * N * 2);
#pragma omp parallel
    {
        REAL *ws = (REAL *)malloc(sizeof(REAL) * per);
        if (!ws) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
#pragma omp for schedule(dynamic, 1)
            for (int b = 0; b < B; ++b) {
                int len = (int)lengths[b];
                if (len < 1 || len > N - 1) { logZ[b] = (REAL)NAN; continue; }
                FN(dmv1o_one)(dec + (size_t)b * N * 8, attach + (size_t)b * N * N * 2, len, N, semiring, (REAL)neg_inf,
                              glogZ ? glogZ[b] : (REAL)1, logZ + b, gdec ? gdec + (size_t)b * N * 8 : 0,
                              gatt ? gatt + (size_t)b * N * N * 2 : 0, ws);
            }
        }
        free(ws);
    }
    return fail ? -1 : 0;
}

#undef C_
#undef I_
#undef gC_
#undef gI_
#undef CL
#undef CR
#undef IL
#undef IR
#undef gCL
#undef gCR
#undef gIL
#undef gIR

/* ------------------------------------------------------------------------------------------
 * DepTree (first-order projective, single root) inside + outside for ONE sentence.
 * deptree.py:25-76.  Charts I, C are NxN:  C[h][e] complete span head h reaching e (either
 * side), I[h][c] incomplete span with arc h -> c.  _check_potentials (deptree.py:146-162)
 * overwrites rows / columns beyond the sentence with the semiring zero on a clone.
 * ------------------------------------------------------------------------------------------ */
#define Cd(h, e) Cc[(h) * N + (e)]
#define Id(h, c) Ic[(h) * N + (c)]
#define gCd(h, e) gC[(h) * N + (e)]
#define gId(h, c) gI[(h) * N + (c)]

static void FN(deptree_one)(const REAL *arc_in, int len, int N, int semiring, REAL neg_inf, REAL glogZ, REAL *logZ,
                            REAL *garc, REAL *ws) {
    const int M = N * N;
    REAL *Cc = ws, *Ic = ws + M, *gC = ws + 2 * M, *gI = ws + 3 * M, *arc = ws + 4 * M, *T = ws + 5 * M;
    REAL *t = T + M, *wg = t + N;
    for (int h = 0; h < N; ++h)                                             /* deptree.py:159-161 */
        for (int c = 0; c < N; ++c) arc[h * N + c] = (h > len || c > len) ? neg_inf : arc_in[h * N + c];
    for (int x = 0; x < M; ++x) { Cc[x] = neg_inf; Ic[x] = neg_inf; }       /* deptree.py:42-43 */
    for (int i = 0; i < N; ++i) Cd(i, i) = 0;                               /* deptree.py:44 */
    for (int w = 1; w < N; ++w) {
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            for (int r = 0; r < w; ++r) t[r] = Cd(i, i + r) + Cd(j, i + r + 1);      /* :53-54 */
            REAL s = FN(sr_sum)(t, w, semiring, 0);
            T[i * N + j] = s;
            Id(j, i) = s + arc[j * N + i];                                           /* :58 */
            Id(i, j) = s + arc[i * N + j];                                           /* :62 */
        }
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            for (int r = 0; r < w; ++r) t[r] = Cd(i + r, i) + Id(j, i + r);          /* :65-66 */
            Cd(j, i) = FN(sr_sum)(t, w, semiring, 0);
            for (int r = 0; r < w; ++r) t[r] = Id(i, i + 1 + r) + Cd(i + 1 + r, j);  /* :68-69 */
            Cd(i, j) = FN(sr_sum)(t, w, semiring, 0);
        }
        if (len != w) Cd(0, w) = neg_inf;                                            /* :71-72 */
    }
    *logZ = Cd(0, len);                                                              /* :74-75 */
    if (!garc) return;

    for (int x = 0; x < M; ++x) { gC[x] = 0; gI[x] = 0; }
    gCd(0, len) = glogZ;
    for (int w = N - 1; w >= 1; --w) {
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            REAL g = gCd(j, i);
            if (g != 0) {
                for (int r = 0; r < w; ++r) t[r] = Cd(i + r, i) + Id(j, i + r);
                FN(sr_sum)(t, w, semiring, wg);
                for (int r = 0; r < w; ++r) { gCd(i + r, i) += g * wg[r]; gId(j, i + r) += g * wg[r]; }
            }
            g = (i == 0 && len != w) ? (REAL)0 : gCd(i, j);
            if (g != 0) {
                for (int r = 0; r < w; ++r) t[r] = Id(i, i + 1 + r) + Cd(i + 1 + r, j);
                FN(sr_sum)(t, w, semiring, wg);
                for (int r = 0; r < w; ++r) { gId(i, i + 1 + r) += g * wg[r]; gCd(i + 1 + r, j) += g * wg[r]; }
            }
        }
        for (int k = 0; k < N - w; ++k) {
            const int i = k, j = k + w;
            REAL gl = gId(j, i), gr = gId(i, j);
            /* masked potentials are overwritten copies: their gradient does not reach the input */
            if (!(j > len)) { garc[j * N + i] += gl; garc[i * N + j] += gr; }
            REAL gs = gl + gr;
            if (gs != 0) {
                for (int r = 0; r < w; ++r) t[r] = Cd(i, i + r) + Cd(j, i + r + 1);
                FN(sr_sum)(t, w, semiring, wg);
                for (int r = 0; r < w; ++r) { gCd(i, i + r) += gs * wg[r]; gCd(j, i + r + 1) += gs * wg[r]; }
            }
        }
    }
}

int FN(orc_deptree)(const REAL *arc, const long long *lengths, int B, int N, int semiring, double neg_inf,
                    const REAL *glogZ, REAL *logZ, REAL *garc) {
    if (B < 0 || N < 2) return -1;
    const size_t per = (size_t)6 * N * N + 2 * (size_t)N;
    int fail = 0;
    if (garc) memset(garc, 0, sizeof(REAL) * (size_t)B * N * N);
#pragma omp parallel
    {
        REAL *ws = (REAL *)malloc(sizeof(REAL) * per);
        if (!ws) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
#pragma omp for schedule(dynamic, 1)
            for (int b = 0; b < B; ++b) {
                int len = lengths ? (int)lengths[b] : N - 1;
                if (len < 1 || len > N - 1) { logZ[b] = (REAL)NAN; continue; }
                FN(deptree_one)(arc + (size_t)b * N * N, len, N, semiring, (REAL)neg_inf, glogZ ? glogZ[b] : (REAL)1,
                                logZ + b, garc ? garc + (size_t)b * N * N : 0, ws);
            }
        }
        free(ws);
    }
    return fail ? -1 : 0;
}

#undef Cd
#undef Id
#undef gCd
#undef gId

/* ------------------------------------------------------------------------------------------
 * Region x word bilinear alignment, src/model/joint.py:406-419 (gather_logit_simple):
 *   attmap[b,a,q,v] = sum_d txt[b,q,d] * vis[a,v,d];  = neg_inf where !vmask[a,v] or !tmask[b,q]
 * Optional fused reductions (consumers: joint.py:473-483, :519-524):
 *   maxV[b,a,q] = max_v attmap ; maxQ[b,a,v] = max_q attmap ; diag[b,q,v] = attmap[b,b,q,v] (A==B)
 * ------------------------------------------------------------------------------------------ */
int FN(orc_bilinear_align)(const REAL *txt, const REAL *vis, const unsigned char *tmask, const unsigned char *vmask,
                           int B, int A, int Q, int V, int d, double neg_inf, REAL *out_full, REAL *out_maxV,
                           REAL *out_maxQ, REAL *out_diag) {
    if (out_diag && A != B) return -1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            for (int v = 0; v < V && out_maxQ; ++v) out_maxQ[((size_t)b * A + a) * V + v] = (REAL)-INFINITY;
            for (int q = 0; q < Q; ++q) {
                REAL mv = (REAL)-INFINITY;
                for (int v = 0; v < V; ++v) {
                    REAL acc = 0;
                    const REAL *x = txt + ((size_t)b * Q + q) * d, *y = vis + ((size_t)a * V + v) * d;
                    for (int k = 0; k < d; ++k) acc += x[k] * y[k];
                    if ((tmask && !tmask[(size_t)b * Q + q]) || (vmask && !vmask[(size_t)a * V + v])) acc = (REAL)neg_inf;
                    if (out_full) out_full[(((size_t)b * A + a) * Q + q) * V + v] = acc;
                    if (acc > mv) mv = acc;
                    if (out_maxQ) { REAL *p = out_maxQ + ((size_t)b * A + a) * V + v; if (acc > *p) *p = acc; }
                    if (out_diag && a == b) out_diag[((size_t)b * Q + q) * V + v] = acc;
                }
                if (out_maxV) out_maxV[((size_t)b * A + a) * Q + q] = mv;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Adjoint of the alignment above for a cotangent g [B,A,Q,V] of attmap -- what autograd derives for
 * joint.py:413-418 (einsum, then two masked_fill_: the filled positions pass no gradient):
 *   g_txt[b,q,:] = sum_{a,v} keep(b,a,q,v) g[b,a,q,v] vis[a,v,:]     g_vis[a,v,:] = sum_{b,q} keep(b,a,q,v) g[b,a,q,v] txt[b,q,:]
 *   keep = tmask[b,q] && vmask[a,v]
 * ------------------------------------------------------------------------------------------ */
int FN(orc_bilinear_align_bwd)(const REAL *g, const REAL *txt, const REAL *vis, const unsigned char *tmask,
                               const unsigned char *vmask, int B, int A, int Q, int V, int d, REAL *g_txt, REAL *g_vis) {
    if (g_txt) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < Q; ++q) {
                REAL *o = g_txt + ((size_t)b * Q + q) * d;
                for (int k = 0; k < d; ++k) o[k] = 0;
                if (tmask && !tmask[(size_t)b * Q + q]) continue;
                for (int a = 0; a < A; ++a)
                    for (int v = 0; v < V; ++v) {
                        if (vmask && !vmask[(size_t)a * V + v]) continue;
                        const REAL w = g[(((size_t)b * A + a) * Q + q) * V + v];
                        const REAL *y = vis + ((size_t)a * V + v) * d;
                        for (int k = 0; k < d; ++k) o[k] += w * y[k];
                    }
            }
    }
    if (g_vis) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int a = 0; a < A; ++a)
            for (int v = 0; v < V; ++v) {
                REAL *o = g_vis + ((size_t)a * V + v) * d;
                for (int k = 0; k < d; ++k) o[k] = 0;
                if (vmask && !vmask[(size_t)a * V + v]) continue;
                for (int b = 0; b < B; ++b)
                    for (int q = 0; q < Q; ++q) {
                        if (tmask && !tmask[(size_t)b * Q + q]) continue;
                        const REAL w = g[(((size_t)b * A + a) * Q + q) * V + v];
                        const REAL *x = txt + ((size_t)b * Q + q) * d;
                        for (int k = 0; k < d; ++k) o[k] += w * x[k];
                    }
            }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Attention-fuse feeding the parser, src/model/joint.py:670-674:
 *   att[b,q,:] = softmax_v( sum_d vis[b,v,d] * txt[b,1+q,d] )          (no region masking: faithful)
 *   x[b,q,:]   = sum_v att[b,q,v] * vis_mid[b,v,:]
 *   out[b,q,:] = LayerNorm_h(enc_x[b,q,:] + x[b,q,:]) * gamma + beta   (biased variance, eps inside sqrt)
 * txt has Lq+1 rows per sentence (root slot first, skipped).
 * ------------------------------------------------------------------------------------------ */
int FN(orc_attn_fuse)(const REAL *vis, const REAL *txt, const REAL *vis_mid, const REAL *enc_x, const REAL *gamma,
                      const REAL *beta, int B, int Lq, int V, int d, int h, double eps, REAL *out_att, REAL *out) {
    int fail = 0;
#pragma omp parallel
    {
        REAL *s = (REAL *)malloc(sizeof(REAL) * (size_t)(V + h));
        if (!s) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
            REAL *y = s + V;
#pragma omp for collapse(2) schedule(static)
            for (int b = 0; b < B; ++b)
                for (int q = 0; q < Lq; ++q) {
                    const REAL *x = txt + ((size_t)b * (Lq + 1) + 1 + q) * d;
                    REAL m = (REAL)-INFINITY;
                    for (int v = 0; v < V; ++v) {
                        const REAL *r = vis + ((size_t)b * V + v) * d;
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += r[k] * x[k];
                        s[v] = acc;
                        if (acc > m) m = acc;
                    }
                    REAL z = 0;
                    for (int v = 0; v < V; ++v) { s[v] = EXP(s[v] - m); z += s[v]; }
                    for (int v = 0; v < V; ++v) {
                        s[v] /= z;
                        if (out_att) out_att[((size_t)b * Lq + q) * V + v] = s[v];
                    }
                    REAL mean = 0;
                    for (int c = 0; c < h; ++c) {
                        REAL acc = 0;
                        for (int v = 0; v < V; ++v) acc += s[v] * vis_mid[((size_t)b * V + v) * h + c];
                        y[c] = enc_x[((size_t)b * Lq + q) * h + c] + acc;
                        mean += y[c];
                    }
                    mean /= h;
                    REAL var = 0;
                    for (int c = 0; c < h; ++c) var += (y[c] - mean) * (y[c] - mean);
                    var /= h;
                    REAL rstd = (REAL)1 / SQRT(var + (REAL)eps);
                    for (int c = 0; c < h; ++c)
                        out[((size_t)b * Lq + q) * h + c] = (y[c] - mean) * rstd * gamma[c] + beta[c];
                }
        }
        free(s);
    }
    return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * Adjoint of the attention-fuse above (what autograd derives for joint.py:670-674), by hand:
 *   y = enc_x + M,  M = P . mid,  P = softmax_v(S),  S = t . vis^T,  out = yhat * gamma + beta,  yhat = (y - mean) * rstd
 *   d_beta  = sum_{b,q} dout                       d_gamma = sum_{b,q} dout * yhat
 *   dyhat   = dout * gamma
 *   dy      = rstd * (dyhat - mean_c(dyhat) - yhat * mean_c(dyhat * yhat))         (LayerNorm, biased variance)
 *   d_enc_x = dy
 *   dP[q,v] = sum_c dy[q,c] mid[v,c]               d_mid[v,c] = sum_q P[q,v] dy[q,c]
 *   dS      = P o (dP - sum_v P o dP)                                               (softmax)
 *   d_txt[1+q,:] = sum_v dS[q,v] vis[v,:]          d_vis[v,:] = sum_q dS[q,v] txt[1+q,:]      d_txt[0,:] = 0
 * One thread per sentence (the sums over words stay inside a sentence); d_gamma / d_beta are reduced over
 * sentences serially afterwards so the result does not depend on the thread count.
 * ------------------------------------------------------------------------------------------ */
int FN(orc_attn_fuse_bwd)(const REAL *vis, const REAL *txt, const REAL *vis_mid, const REAL *enc_x, const REAL *gamma,
                          const REAL *dout, int B, int Lq, int V, int d, int h, double eps, REAL *d_vis, REAL *d_txt,
                          REAL *d_mid, REAL *d_enc, REAL *d_gamma, REAL *d_beta) {
    int fail = 0;
    REAL *part = (REAL *)calloc((size_t)B * 2 * h + 1, sizeof(REAL));   /* per-sentence d_gamma / d_beta */
    if (!part) return -1;
#pragma omp parallel
    {
        REAL *s = (REAL *)malloc(sizeof(REAL) * (size_t)(2 * V + 3 * h));
        if (!s) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
            REAL *dp = s + V, *y = dp + V, *yh = y + h, *dy = yh + h;
#pragma omp for schedule(static)
            for (int b = 0; b < B; ++b) {
                REAL *dv = d_vis + (size_t)b * V * d, *dm = d_mid + (size_t)b * V * h;
                for (size_t i = 0; i < (size_t)V * d; ++i) dv[i] = 0;
                for (size_t i = 0; i < (size_t)V * h; ++i) dm[i] = 0;
                for (int k = 0; k < d; ++k) d_txt[(size_t)b * (Lq + 1) * d + k] = 0;
                for (int q = 0; q < Lq; ++q) {
                    const REAL *x = txt + ((size_t)b * (Lq + 1) + 1 + q) * d;
                    REAL m = (REAL)-INFINITY;
                    for (int v = 0; v < V; ++v) {
                        const REAL *r = vis + ((size_t)b * V + v) * d;
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += r[k] * x[k];
                        s[v] = acc;
                        if (acc > m) m = acc;
                    }
                    REAL z = 0;
                    for (int v = 0; v < V; ++v) { s[v] = EXP(s[v] - m); z += s[v]; }
                    for (int v = 0; v < V; ++v) s[v] /= z;
                    REAL mean = 0;
                    for (int c = 0; c < h; ++c) {
                        REAL acc = 0;
                        for (int v = 0; v < V; ++v) acc += s[v] * vis_mid[((size_t)b * V + v) * h + c];
                        y[c] = enc_x[((size_t)b * Lq + q) * h + c] + acc;
                        mean += y[c];
                    }
                    mean /= h;
                    REAL var = 0;
                    for (int c = 0; c < h; ++c) var += (y[c] - mean) * (y[c] - mean);
                    var /= h;
                    const REAL rstd = (REAL)1 / SQRT(var + (REAL)eps);
                    const REAL *go = dout + ((size_t)b * Lq + q) * h;
                    REAL c1 = 0, c2 = 0;
                    for (int c = 0; c < h; ++c) {
                        yh[c] = (y[c] - mean) * rstd;
                        part[((size_t)b * 2) * h + c] += go[c] * yh[c];
                        part[((size_t)b * 2 + 1) * h + c] += go[c];
                        const REAL dyh = go[c] * gamma[c];
                        c1 += dyh;
                        c2 += dyh * yh[c];
                    }
                    c1 /= h;
                    c2 /= h;
                    for (int c = 0; c < h; ++c) {
                        dy[c] = rstd * (go[c] * gamma[c] - c1 - yh[c] * c2);
                        d_enc[((size_t)b * Lq + q) * h + c] = dy[c];
                    }
                    REAL dot = 0;
                    for (int v = 0; v < V; ++v) {
                        const REAL *mr = vis_mid + ((size_t)b * V + v) * h;
                        REAL acc = 0;
                        for (int c = 0; c < h; ++c) { acc += dy[c] * mr[c]; dm[(size_t)v * h + c] += s[v] * dy[c]; }
                        dp[v] = acc;
                        dot += s[v] * acc;
                    }
                    REAL *dx = d_txt + ((size_t)b * (Lq + 1) + 1 + q) * d;
                    for (int k = 0; k < d; ++k) dx[k] = 0;
                    for (int v = 0; v < V; ++v) {
                        const REAL ds = s[v] * (dp[v] - dot);
                        const REAL *r = vis + ((size_t)b * V + v) * d;
                        for (int k = 0; k < d; ++k) { dx[k] += ds * r[k]; dv[(size_t)v * d + k] += ds * x[k]; }
                    }
                }
            }
        }
        free(s);
    }
    if (!fail) {
        for (int c = 0; c < h; ++c) { d_gamma[c] = 0; d_beta[c] = 0; }
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < h; ++c) {
                d_gamma[c] += part[((size_t)b * 2) * h + c];
                d_beta[c] += part[((size_t)b * 2 + 1) * h + c];
            }
    }
    free(part);
    return fail ? -1 : 0;
}

/* ------------------------------------------------------------------------------------------
 * Grounding loss on the alignment, src/model/joint.py:406-419 (gather_logit_simple) + :439-491
 * (loss_grounding_factor_ce), and its gradient to both feature tensors.  A == B (caption b pairs with image b).
 *   att[b,a,q,v] = <txt[b,q], vis[a,v]>, masked entries = neg_inf (joint.py:417-418); on the diagonal pairs a == b the
 *   POS prior subtracts pen[b,q,seg(v)] (joint.py:446-470; the caller builds pen from the tags: 100 for every named
 *   factor whose POS set holds the token's tag and whose segment is not seg(v); rows q outside 1..L carry 0).
 *   mV[b,a,q] = max_v att   ->  txt2vis = - sum_{b,q} marg[b,q] * log_softmax_a(mV)[b,b,q]        (joint.py:472-476)
 *   mQ[b,a,v] = max_q att   ->  vis2txt = - sum_{a,v} vmask[a,v] * log_softmax_b(mQ)[a,a,v]       (joint.py:478-483)
 *   total = txt2vis / (txt2vis + 1e-6) * num  +  w * vis2txt / (vis2txt + 1e-6) * num  (denominators detached)
 * Gradient: d total / d txt2vis = num / (txt2vis + 1e-6) etc.; through log_softmax (p - delta), through max (to the
 * first arg-max), through masked_fill (nothing where either mask is off), through the contraction.
 * out_sums = {txt2vis, vis2txt, total}.  Optional outputs may be NULL.
 * ------------------------------------------------------------------------------------------ */
int FN(orc_grounding_loss)(const REAL *txt, const REAL *vis, const uint8_t *tmask, const uint8_t *vmask, const REAL *marg,
                           const REAL *pen, const uint8_t *seg_of_v, int n_seg, int B, int Q, int V, int d,
                           double neg_inf, double num_token, double w_v2t, REAL *out_sums, REAL *out_maxV,
                           int32_t *out_argV, REAL *out_maxQ, int32_t *out_argQ, REAL *g_txt, REAL *g_vis) {
    const int A = B;
    REAL *mV = (REAL *)malloc(sizeof(REAL) * (size_t)B * A * Q), *mQ = (REAL *)malloc(sizeof(REAL) * (size_t)B * A * V);
    int32_t *aV = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * A * Q), *aQ = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * A * V);
    if (!mV || !mQ || !aV || !aQ) { free(mV); free(mQ); free(aV); free(aQ); return -1; }
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int a = 0; a < A; ++a) {
            REAL *mv = mV + ((size_t)b * A + a) * Q, *mq = mQ + ((size_t)b * A + a) * V;
            int32_t *av = aV + ((size_t)b * A + a) * Q, *aq = aQ + ((size_t)b * A + a) * V;
            for (int v = 0; v < V; ++v) { mq[v] = (REAL)-INFINITY; aq[v] = 0; }
            for (int q = 0; q < Q; ++q) {
                const REAL *x = txt + ((size_t)b * Q + q) * d;
                REAL best = (REAL)-INFINITY;
                int bi = 0;
                for (int v = 0; v < V; ++v) {
                    REAL val;
                    if ((tmask && !tmask[(size_t)b * Q + q]) || (vmask && !vmask[(size_t)a * V + v])) val = (REAL)neg_inf;
                    else {
                        const REAL *r = vis + ((size_t)a * V + v) * d;
                        REAL acc = 0;
                        for (int k = 0; k < d; ++k) acc += x[k] * r[k];
                        val = acc;
                    }
                    if (pen && a == b) val -= pen[((size_t)b * Q + q) * n_seg + seg_of_v[v]];
                    if (val > best) { best = val; bi = v; }
                    if (val > mq[v]) { mq[v] = val; aq[v] = q; }
                }
                mv[q] = best;
                av[q] = bi;
            }
        }
    /* the two cross-entropies */
    double t2v = 0, v2t = 0;
    for (int b = 0; b < B; ++b)
        for (int q = 0; q < Q; ++q) {
            REAL m = (REAL)-INFINITY, z = 0;
            for (int a = 0; a < A; ++a) if (mV[((size_t)b * A + a) * Q + q] > m) m = mV[((size_t)b * A + a) * Q + q];
            for (int a = 0; a < A; ++a) z += EXP(mV[((size_t)b * A + a) * Q + q] - m);
            t2v -= (double)marg[(size_t)b * Q + q] * (double)((mV[((size_t)b * A + b) * Q + q] - m) - LOG(z));
        }
    for (int a = 0; a < A; ++a)
        for (int v = 0; v < V; ++v) {
            REAL m = (REAL)-INFINITY, z = 0;
            for (int b = 0; b < B; ++b) if (mQ[((size_t)b * A + a) * V + v] > m) m = mQ[((size_t)b * A + a) * V + v];
            for (int b = 0; b < B; ++b) z += EXP(mQ[((size_t)b * A + a) * V + v] - m);
            const double keep = vmask ? (vmask[(size_t)a * V + v] ? 1.0 : 0.0) : 1.0;
            v2t -= keep * (double)((mQ[((size_t)a * A + a) * V + v] - m) - LOG(z));
        }
    const double c1 = num_token / (t2v + 1e-6), c2 = w_v2t > 0 ? w_v2t * num_token / (v2t + 1e-6) : 0.0;
    if (out_sums) {
        out_sums[0] = (REAL)t2v;
        out_sums[1] = (REAL)v2t;
        out_sums[2] = (REAL)(t2v * c1 + v2t * c2);
    }
    if (out_maxV) memcpy(out_maxV, mV, sizeof(REAL) * (size_t)B * A * Q);
    if (out_argV) memcpy(out_argV, aV, sizeof(int32_t) * (size_t)B * A * Q);
    if (out_maxQ) memcpy(out_maxQ, mQ, sizeof(REAL) * (size_t)B * A * V);
    if (out_argQ) memcpy(out_argQ, aQ, sizeof(int32_t) * (size_t)B * A * V);
    if (g_txt && g_vis) {
        for (size_t i = 0; i < (size_t)B * Q * d; ++i) g_txt[i] = 0;
        for (size_t i = 0; i < (size_t)A * V * d; ++i) g_vis[i] = 0;
        /* serial on purpose: the scatter into g_vis / g_txt has a fixed order */
        for (int b = 0; b < B; ++b)
            for (int q = 0; q < Q; ++q) {
                if (tmask && !tmask[(size_t)b * Q + q]) continue;
                REAL m = (REAL)-INFINITY, z = 0;
                for (int a = 0; a < A; ++a) if (mV[((size_t)b * A + a) * Q + q] > m) m = mV[((size_t)b * A + a) * Q + q];
                for (int a = 0; a < A; ++a) z += EXP(mV[((size_t)b * A + a) * Q + q] - m);
                for (int a = 0; a < A; ++a) {
                    const int v = aV[((size_t)b * A + a) * Q + q];
                    if (vmask && !vmask[(size_t)a * V + v]) continue;
                    const REAL p = EXP(mV[((size_t)b * A + a) * Q + q] - m) / z;
                    const REAL g = (REAL)c1 * marg[(size_t)b * Q + q] * (p - (a == b ? (REAL)1 : (REAL)0));
                    REAL *gt = g_txt + ((size_t)b * Q + q) * d, *gv = g_vis + ((size_t)a * V + v) * d;
                    const REAL *x = txt + ((size_t)b * Q + q) * d, *r = vis + ((size_t)a * V + v) * d;
                    for (int k = 0; k < d; ++k) { gt[k] += g * r[k]; gv[k] += g * x[k]; }
                }
            }
        if (c2 != 0.0)
            for (int a = 0; a < A; ++a)
                for (int v = 0; v < V; ++v) {
                    if (vmask && !vmask[(size_t)a * V + v]) continue;
                    REAL m = (REAL)-INFINITY, z = 0;
                    for (int b = 0; b < B; ++b) if (mQ[((size_t)b * A + a) * V + v] > m) m = mQ[((size_t)b * A + a) * V + v];
                    for (int b = 0; b < B; ++b) z += EXP(mQ[((size_t)b * A + a) * V + v] - m);
                    for (int b = 0; b < B; ++b) {
                        const int q = aQ[((size_t)b * A + a) * V + v];
                        if (tmask && !tmask[(size_t)b * Q + q]) continue;
                        const REAL p = EXP(mQ[((size_t)b * A + a) * V + v] - m) / z;
                        const REAL g = (REAL)c2 * (p - (a == b ? (REAL)1 : (REAL)0));
                        REAL *gt = g_txt + ((size_t)b * Q + q) * d, *gv = g_vis + ((size_t)a * V + v) * d;
                        const REAL *x = txt + ((size_t)b * Q + q) * d, *r = vis + ((size_t)a * V + v) * d;
                        for (int k = 0; k < d; ++k) { gt[k] += g * r[k]; gv[k] += g * x[k]; }
                    }
                }
    }
    free(mV); free(mQ); free(aV); free(aQ);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Arc encoder of lang_feat word+maxdep, src/model/joint.py:281-287:
 *   arc[m,h] = sum_{x,y} child[m,x] * w1[x,h,y] * parent[m,y]  +  sum_x (child + parent)[m,x] * w2[x,h]  +  b[h]
 * (m runs over batch x positions) and its adjoint for the cotangent g[m,h]:
 *   d_child[m,x] = sum_{h,y} g w1 parent + sum_h g[m,h] w2[x,h]          d_parent likewise (w1 contracted over x, h)
 *   d_w1[x,h,y]  = sum_m child[m,x] g[m,h] parent[m,y]     d_w2[x,h] = sum_m (child + parent)[m,x] g[m,h]     d_b = sum_m g
 * w2 / b may be NULL (trilinear term only).  Gradient outputs may be NULL individually.
 * ------------------------------------------------------------------------------------------ */
int FN(orc_arc_encoder)(const REAL *child, const REAL *parent, const REAL *w1, const REAL *w2, const REAL *b, int M,
                        int X, int H, int Y, REAL *out) {
#pragma omp parallel for schedule(static)
    for (int m = 0; m < M; ++m)
        for (int h = 0; h < H; ++h) {
            REAL acc = b ? b[h] : (REAL)0;
            for (int x = 0; x < X; ++x) {
                const REAL *wr = w1 + ((size_t)x * H + h) * Y;
                REAL t = 0;
                for (int y = 0; y < Y; ++y) t += wr[y] * parent[(size_t)m * Y + y];
                acc += child[(size_t)m * X + x] * t;
                if (w2) acc += (child[(size_t)m * X + x] + parent[(size_t)m * Y + x]) * w2[(size_t)x * H + h];
            }
            out[(size_t)m * H + h] = acc;
        }
    return 0;
}

int FN(orc_arc_encoder_bwd)(const REAL *child, const REAL *parent, const REAL *w1, const REAL *w2, const REAL *g, int M, int X,
                            int H, int Y, REAL *d_child, REAL *d_parent, REAL *d_w1, REAL *d_w2, REAL *d_b) {
    if (d_child || d_parent) {
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m) {
            if (d_child)
                for (int x = 0; x < X; ++x) {
                    REAL acc = 0;
                    for (int h = 0; h < H; ++h) {
                        const REAL *wr = w1 + ((size_t)x * H + h) * Y;
                        REAL t = 0;
                        for (int y = 0; y < Y; ++y) t += wr[y] * parent[(size_t)m * Y + y];
                        acc += g[(size_t)m * H + h] * (t + (w2 ? w2[(size_t)x * H + h] : (REAL)0));
                    }
                    d_child[(size_t)m * X + x] = acc;
                }
            if (d_parent)
                for (int y = 0; y < Y; ++y) {
                    REAL acc = 0;
                    for (int h = 0; h < H; ++h) {
                        REAL t = 0;
                        for (int x = 0; x < X; ++x) t += child[(size_t)m * X + x] * w1[((size_t)x * H + h) * Y + y];
                        acc += g[(size_t)m * H + h] * (t + (w2 ? w2[(size_t)y * H + h] : (REAL)0));
                    }
                    d_parent[(size_t)m * Y + y] = acc;
                }
        }
    }
    if (d_w1) {
#pragma omp parallel for schedule(static)
        for (int x = 0; x < X; ++x)
            for (int h = 0; h < H; ++h)
                for (int y = 0; y < Y; ++y) {
                    REAL acc = 0;
                    for (int m = 0; m < M; ++m) acc += child[(size_t)m * X + x] * g[(size_t)m * H + h] * parent[(size_t)m * Y + y];
                    d_w1[((size_t)x * H + h) * Y + y] = acc;
                }
    }
    if (d_w2)
        for (int x = 0; x < X; ++x)
            for (int h = 0; h < H; ++h) {
                REAL acc = 0;
                for (int m = 0; m < M; ++m) acc += (child[(size_t)m * X + x] + parent[(size_t)m * Y + x]) * g[(size_t)m * H + h];
                d_w2[(size_t)x * H + h] = acc;
            }
    if (d_b)
        for (int h = 0; h < H; ++h) {
            REAL acc = 0;
            for (int m = 0; m < M; ++m) acc += g[(size_t)m * H + h];
            d_b[h] = acc;
        }
    return 0;
}

#undef DEC
#undef ATT
#undef GDEC
#undef GATT
#undef FN
#undef CAT
#undef CAT_
