"""The parser's feed-forwards in front of the score construction -- host-side mirror of `DiscriminativeNDMV._forward`,
src/model/ldndmv.py:174-205 (context_mode 'mean'), from the embeddings to the scorers' projected inputs:

    h        = cat([emb, mean_l(x)])                                    :174-177, extract_sent_repr :226, construct_token_repr :254
    h_parent = mid_ff(head_ff(h))        [B,L,2,2,H]                    :180   (MLP nn/common.py:23-51; DMVSkipConnectEncoder nn/dmv_spec.py:6-54)
    h_child  = mid_ff(child_ff(token_emb)), h_root = mid_ff(root_ff(root_emb)), h_dec = mid_ff(dec_ff(dec_emb))     :181-183
    x1, x2   = attach_scorer.project1(h_parent), .project2(h_child)      nn/dmv_spec.py:67-68
    y1, y2   = dec_scorer.project1(h_parent),    .project2(h_dec)
    root_rule = root_scorer(h_root, h_child).sum([-1,-2]).log_softmax(-1)                                           :205

These are plain Linear / LeakyReLU stacks: the GEMMs over the ~10^4 token rows stay with the library (rocBLAS / hipBLASLt).  What this module changes is how
many of them there are and what surrounds them -- as the reference's modules run them (5-D inputs, one nn.Linear at a time, autograd)
the stage is ~190 launches and 5 ms of device time at B = 256, L = 40 in bf16, most of it weight-gradient GEMMs on four workgroups,
bias-gradient reductions and un-fused bias / residual adds.  Here, with identical mathematics:
  * ONE pass of mid_ff over the rows of all four inputs (B L + T + 1 + 2 rows);
  * every Linear on a 2-D view with its bias inside the GEMM (addmm);
  * each bottleneck pair Linear(H, nb) -> Linear(nb, H) (no activation between them, nn/dmv_spec.py:52-54) folded into one H x H
    weight, W1 W0, and HASCHILD | NOCHILD (LEFT | RIGHT) concatenated into one [2H, H] GEMM; the adjoint unfolds the pair;
  * linear2 folded into the six scorer projections that follow it without an activation (h W2^T + b2) P^T + p = h (P W2)^T + (P b2 + p),
    so the [4 B L, H] x [H, H] GEMM of linear2 and its two adjoints never run;
  * head_ff's weight split into its embedding and context columns: the context term is one [B, h] x [h, H] product per sentence
    instead of L copies of it inside a concatenated [B L, E + h] operand;
  * a hand-written backward in the same granularity with every weight / bias gradient on the split-K kernel (vlg_linear_wgrad);
  * everything BETWEEN two GEMMs -- bias / context / residual add, LeakyReLU, dropout mask, the (val, dir) -> (dir, val) stack, the
    skip connections' cotangent sums, LeakyReLU' -- as one pass per stage (csrc/vlg_ff.hip) instead of 2-4 torch launches over a
    [4 B L, H] activation each;
  * the products in WEIGHT space (the fold W1 W0 and its unfolding, the per-sentence context term, the token / root / decision MLPs, the
    root rule, the small projection block) on `align.small_matmul` (vlg_small_gemm: one wavefront per 32 x 32 output tile, operands read
    through their strides) -- the library maps an output of a few hundred rows and columns to one workgroup.
The reference formulation (tools/train_step.scorer_feed_forward, module by module) is what the tests compare this with.
"""
import ctypes
import os

import torch
from torch.autograd.function import once_differentiable

from . import _C
from .align import SmallMatmulGroup, WgradGroup, _wgrad_ok, linear_kn, linear_kn_ok, linear_wgrad, small_matmul

SLOPE = 0.01   # nn.LeakyReLU() default (nn/common.py:31, nn/dmv_spec.py:10)
SITE_MID_FF = 2   # the dropout layer id of mid_ff's nn.Dropout in the step's shared counter-based generator (encoders.SITE_MID_FF)
_BOTTLENECKS = ("NOCHILD_linear", "HASCHILD_linear", "LEFT_linear", "RIGHT_linear")   # stack order: [no, has] (:42), [left, right] (:47)
_PROJ = ("attach_scorer.project1", "dec_scorer.project1", "attach_scorer.project2", "root_scorer.project2", "root_scorer.project1",
         "dec_scorer.project2")      # first two: the rows of h_parent; then h_child (x2, root's second operand), h_root, h_dec


def param_names(n_bottleneck):
    """The reference modules' parameter names (behind "ff.") in the order `parser_feed_forward` takes them."""
    names = []
    for m in ("head_ff", "child_ff", "root_ff", "dec_ff"):
        names += [f"ff.{m}.linear.weight", f"ff.{m}.linear.bias"]
    for b in _BOTTLENECKS:
        if n_bottleneck:
            names += [f"ff.mid_ff.{b}.0.weight", f"ff.mid_ff.{b}.0.bias", f"ff.mid_ff.{b}.1.weight", f"ff.mid_ff.{b}.1.bias"]
        else:
            names += [f"ff.mid_ff.{b}.weight", f"ff.mid_ff.{b}.bias"]
    for m in ("valence_linear", "direction_linear", "linear1", "linear2"):
        names += [f"ff.mid_ff.{m}.weight", f"ff.mid_ff.{m}.bias"]
    for p in _PROJ:
        names += [f"ff.{p}.weight", f"ff.{p}.bias"]
    return names


def _validate_shapes(P, nb, H, n_head_in):
    """The fused pass walks every mid_ff activation as rows of H channels and stacks the six scorer projections into one [6r, H] block:
    the reference's DMVSkipConnectEncoder also accepts n_mid != hidden_size and separate attach / dec / root ranks (nn/dmv_spec.py:6-36,
    57-68), which this formulation does NOT cover -- refuse them instead of reading past a buffer."""
    def want(name, shape):
        got = tuple(P[name].shape)
        if got != tuple(shape):
            raise ValueError(f"parser_feed_forward: {name} is {got}, this fused pass needs {tuple(shape)} "
                             "(n_mid == hidden_size and one rank shared by attach / dec / root scorers, as config/model/vlgae.yaml ships)")
    want("ff.head_ff.linear.weight", (H, n_head_in))
    for m in ("child_ff", "root_ff", "dec_ff"):
        if P[f"ff.{m}.linear.weight"].shape[0] != H:
            want(f"ff.{m}.linear.weight", (H, P[f"ff.{m}.linear.weight"].shape[1]))
    for b in _BOTTLENECKS:
        if nb:
            want(f"ff.mid_ff.{b}.0.weight", (nb, H))
            want(f"ff.mid_ff.{b}.1.weight", (H, nb))
        else:
            want(f"ff.mid_ff.{b}.weight", (H, H))
    for m in ("valence_linear", "direction_linear", "linear1", "linear2"):
        want(f"ff.mid_ff.{m}.weight", (H, H))
        want(f"ff.mid_ff.{m}.bias", (H,))
    r = P[f"ff.{_PROJ[0]}.weight"].shape[0]
    for p_ in _PROJ:
        want(f"ff.{p_}.weight", (r, H))
        want(f"ff.{p_}.bias", (r,))


def _numel(shape):
    n = 1
    for v in shape:
        n *= v
    return n


def _adt(t):
    return _C.BF16 if t.dtype == torch.bfloat16 else _C.F32


def _act(inp, out, M, J, H, residual=None, mask=None, swap=False, mask_scale=1.0, rng=None, p=0.0):
    """out[m,j'] = LeakyReLU(inp[m,j] + residual[m]) * mask[m,j'] * mask_scale in one pass (vlg_ff_act); out may be inp unless swap.
    rng (a DeviceRng) instead of mask: the keep-mask is drawn inside the kernel (rate p, site SITE_MID_FF)."""
    _C.check(_C.lib().vlg_ff_act(_C.ptr(inp), _C.ptr(residual), _C.ptr(mask), float(mask_scale), None if rng is None else _C.ptr(rng.state), SITE_MID_FF,
                                 float(p), _C.ptr(out), M, J, H, int(swap), _adt(inp), SLOPE, _C.stream_of(inp)), "ff_act")
    return out


def _act_bwd(g, act, out, M, J, H, mask=None, total=None, accumulate=False, swap=False, mask_scale=1.0, rng=None, p=0.0):
    """out[m,j'] = LeakyReLU'(act[m,j]) * g[m,j] * mask[m,j] * mask_scale; total [M,H] fp32 (+)= sum_j (vlg_ff_act_backward)."""
    _C.check(_C.lib().vlg_ff_act_backward(_C.ptr(g), _C.ptr(act), _C.ptr(mask), float(mask_scale), None if rng is None else _C.ptr(rng.state),
                                          SITE_MID_FF, float(p), _C.ptr(out), _C.ptr(total), M, J, H, int(swap), int(accumulate), _adt(g), SLOPE,
                                          _C.stream_of(g)), "ff_act_backward")
    return out


_FF_LIBRARY = bool(os.environ.get("VLGAE_FF_LIBRARY"))   # opt-out: every Linear of the skip-connect encoder as library GEMM + element-wise pass


def _fused(act, H):
    """Whether the skip-connect encoder's Linear layers run as the row-streaming product fused with their element-wise pass
    (vlg_ff_linear_act: bf16 storage, exactly 256 channels -- the shipped width); otherwise library GEMM + vlg_ff_act as before."""
    return act == torch.bfloat16 and H == 256 and not _FF_LIBRARY


def _linear_act(x, w, bias, out, nb=1, residual=None, rs=0, om=1, oy=0, mask=None, mask_scale=1.0, rng=None, p=0.0):
    """out[orow] = LeakyReLU(bf16(x[row] @ w[256 y:256 y + 256].T + bias) + residual[row >> rs]) * keep, orow = (row >> rs) om + y oy + (row & ((1 << rs) - 1))
    for the nb column blocks y of w [nb 256, 256]: Linear + skip connection + LeakyReLU (+ dropout) in one launch (vlg_ff_linear_act)."""
    _C.check(_C.lib().vlg_ff_linear_act(_C.ptr(x), x.stride(0), _C.ptr(w), _C.ptr(bias), x.shape[0], nb, _C.ptr(residual), rs, om, oy, _C.ptr(mask),
                                        float(mask_scale), None if rng is None else _C.ptr(rng.state), SITE_MID_FF, float(p), _C.ptr(out), SLOPE,
                                        _C.stream_of(x)), "ff_linear_act")
    return out


def _linear_act_bwd(g, w_t, act, out, J=1, mask=None, mask_scale=1.0, rng=None, p=0.0, total=None, accumulate=False, swap=False, w_kn=False):
    """out = LeakyReLU'(act) * bf16(g @ W) * keep with W given transposed (w_t = W.T contiguous); total [rows / J, 256] fp32 (+)= the group sums
    (vlg_ff_linear_act_backward: the input-gradient product of a layer fused with the adjoint of the element-wise pass in front of it).
    g [rows, 512]: w_t [2,256,256], the transposes of W's two row blocks; g [rows, 32] with w_kn: w_t is W itself, [32, 256]."""
    _C.check(_C.lib().vlg_ff_linear_act_backward(_C.ptr(g), g.stride(0), _C.ptr(w_t), g.shape[1], int(w_kn), g.shape[0], J, _C.ptr(act), _C.ptr(mask), float(mask_scale),
                                                 None if rng is None else _C.ptr(rng.state), SITE_MID_FF, float(p), _C.ptr(out), _C.ptr(total),
                                                 int(swap), int(accumulate), SLOPE, _C.stream_of(g)), "ff_linear_act_backward")
    return out


def _stage(w, out, bias=None, mask=None, mask_scale=1.0, rng=None, p=0.0, act=None, total=None, J=1, swap=False, accumulate=False):
    """A VlgFfStage record (and the tensors it points at, to be kept alive until the launch is enqueued)."""
    st = _C.FfStage()
    pt = lambda t: None if t is None else t.data_ptr()
    st.w, st.bias, st.mask, st.rng, st.out, st.act, st.sum = pt(w), pt(bias), pt(mask), (None if rng is None else rng.state.data_ptr()), pt(out), pt(act), pt(total)
    st.mask_scale, st.p, st.site, st.J, st.swap, st.accumulate = float(mask_scale), float(p), SITE_MID_FF, int(J), int(swap), int(accumulate)
    return st


def _linear_act_chain2(x, s1, s2, backward=False):
    """Two consecutive 256 -> 256 stages (`_stage` records) on the rows of x in ONE launch: stage 2 reads stage 1's stored rows from LDS
    (vlg_ff_linear_act_chain2); both outputs are written."""
    _C.check(_C.lib().vlg_ff_linear_act_chain2(_C.ptr(x), x.stride(0), x.shape[0], int(backward), ctypes.byref(s1), ctypes.byref(s2), SLOPE, _C.stream_of(x)),
             "ff_linear_act_chain2")


def _transpose256(mats, out):
    """out[z] = mats[z].T for up to eight contiguous [256,256] bf16 matrices in one launch (vlg_ff_transpose256)."""
    arr = (ctypes.c_void_p * len(mats))(*(m.data_ptr() for m in mats))
    _C.check(_C.lib().vlg_ff_transpose256(arr, len(mats), _C.ptr(out), _C.stream_of(out)), "ff_transpose256")
    return out


def _mask32(m):
    if m is None:
        return None
    m = m.detach().to(torch.float32).contiguous()
    return m.clone() if m.data_ptr() % 16 else m       # (the kernels read the [B,H] masks 16 bytes at a time)


def _wgrad(dy, x, out=None, dtype=torch.float32, defer=None):
    """(dy^T x [out,in], sum_rows dy [out]): the split-K kernel for bf16 token-row counts, the library otherwise.  Results in `dtype` (the
    parameter's own: the reduction writes it, no cast launch follows), or written into `out` = (d_weight, d_bias) in their own dtype
    (contiguous views of a caller's stack: no cat afterwards either)."""
    if dy.dtype == x.dtype and _wgrad_ok(x.shape[0], dy.shape[1], x.shape[1], dy.dtype):
        return linear_wgrad(dy, x, out=out, out_dtype=dtype, defer=defer)
    dw, db = dy.float().t() @ x.float(), dy.float().sum(0)
    if out is None:
        return dw.to(dtype), db.to(dtype)
    out[0].copy_(dw)
    out[1].copy_(db)
    return out


_ONES = {}


def _ones(n, dtype, device):
    """A cached [n,1] tensor of ones (the column that turns small_matmul's rank-one term into a per-row bias)."""
    key = (n, dtype, device)
    if key not in _ONES:
        _ONES[key] = torch.ones((n, 1), dtype=dtype, device=device)
    return _ONES[key]


class _ParserFF(torch.autograd.Function):
    """(emb [B,L,E], x [B,L,h], token_emb [T,Et], root_emb [1,er], dec_emb [2,ed], *params) -> (x1, x2, y1, y2, root_rule)."""

    @staticmethod
    def forward(ctx, nb, drops, emb, x, token_emb, root_emb, dec_emb, *params):
        names = param_names(nb)
        P = dict(zip(names, (p.detach() for p in params)))
        act = emb.dtype
        B, L, E = emb.shape
        h = x.shape[2]
        T = token_emb.shape[0]
        M0, Ms = B * L, T + 3
        M = M0 + Ms
        dev = emb.device
        c = lambda t: t.to(act)
        Wh, bh = c(P["ff.head_ff.linear.weight"]), c(P["ff.head_ff.linear.bias"])
        H = Wh.shape[0]
        We, Wc = Wh[:, :E], Wh[:, E:]
        _validate_shapes(P, nb, H, E + h)
        emb2 = emb.detach().reshape(M0, E)
        drop_head, drop_small, drop_mid, mid_scale = drops
        mid_rng, mid_mask, p_mid = (drop_mid[0], None, drop_mid[1]) if isinstance(drop_mid, tuple) else (None, drop_mid, 0.0)
        lib, st, adt = _C.lib(), _C.stream_of(emb), _C.BF16 if act == torch.bfloat16 else _C.F32
        # ---- the parameters stacked: the 16 + 12 bottleneck / projection tensors gathered (and cast) by ONE multi-tensor copy ----
        r = P[f"ff.{_PROJ[0]}.weight"].shape[0]
        shapes = ([(4, nb, H), (4, nb), (4, H, nb), (4, H)] if nb else [(4, H, H), (4, H)]) + [(6, r, H), (6, r)]
        flat = torch.empty(sum(-(-_numel(sh) // 128) * 128 for sh in shapes), dtype=act, device=dev)   # (every stack 256-byte aligned)
        stacks, o = [], 0
        for sh in shapes:
            stacks.append(flat[o:o + _numel(sh)].view(sh))
            o += -(-_numel(sh) // 128) * 128
        srcs = [[P[f"ff.mid_ff.{b}{sfx}"] for b in _BOTTLENECKS] for sfx in ((".0.weight", ".0.bias", ".1.weight", ".1.bias") if nb else (".weight", ".bias"))]
        srcs += [[P[f"ff.{p}.weight"] for p in _PROJ], [P[f"ff.{p}.bias"] for p in _PROJ]]
        torch._foreach_copy_([st_[k] for st_, src in zip(stacks, srcs) for k in range(len(src))], [t for src in srcs for t in src])
        PW, Pb = stacks[-2].view(6 * r, H), stacks[-1].view(6 * r)
        W2_, b2_ = c(P["ff.mid_ff.linear2.weight"]), c(P["ff.mid_ff.linear2.bias"])
        # ---- MLPs: all rows into one [M, H] buffer ----
        X = torch.empty((M, H), dtype=act, device=dev)
        torch.mm(emb2, We.t(), out=X[:M0])
        xd = x.detach()
        if xd.dtype in (torch.float32, torch.bfloat16) and xd.is_contiguous():           # context_mode 'mean', ldndmv.py:226: cast + mean in one launch
            cmean = torch.empty((B, h), dtype=act, device=dev)
            _C.check(lib.vlg_ff_context_mean(_C.ptr(xd), _adt(xd), B, L, h, _C.ptr(cmean), adt, st), "ff_context_mean")
        else:
            cmean = xd.mean(1, dtype=act)
        # ---- every product in WEIGHT space that depends on parameters (and the sentence means) only, as ONE grouped launch: the context term,
        # the token / root / decision MLP rows, the folded bottlenecks W1 W0 (+ biases), the folded projections P W2 (+ biases) ----
        grp = SmallMatmulGroup()
        cterm = grp.add(cmean, Wc.t(), bias=bh)                                          # [B,H]: the context columns + bias, once per sentence
        small_in = (token_emb, root_emb, dec_emb)
        o = M0
        for m, inp in zip(("child_ff", "root_ff", "dec_ff"), small_in):
            n = inp.shape[0]
            grp.add(c(inp.detach()), c(P[f"ff.{m}.linear.weight"]).t(), bias=c(P[f"ff.{m}.linear.bias"]), out=X[o:o + n])   # (T | 1 | 2 rows)
            o += n
        if nb:
            W0s, b0s, W1s, b1s = stacks[:4]                                                 # [4,nb,H], [4,nb], [4,H,nb], [4,H]
            Weff = grp.add(W1s, W0s)                                                        # [4,H,H] (one wavefront per 32 x 32 tile: the library runs this on 4 workgroups)
            # (the library's batched bf16 matrix-VECTOR product costs ~10 ms of HOST time per call on this stack; here: one column, b1 as the rank-one term)
            beff = grp.add(W1s, b0s.unsqueeze(-1), rank1=(b1s, _ones(4, act, dev))).squeeze(-1)   # [4,H] = W1 b0 + b1
        else:
            W0s = b0s = W1s = b1s = None
            Weff, beff = stacks[:2]
        Wp = grp.add(PW, W2_)                                                            # [6r,H]: P W2 (linear2 folded into the projections)
        bp = grp.add(PW, b2_.unsqueeze(-1), rank1=(Pb, _ones(1, act, dev)[0])).squeeze(-1)   # P b2 + p
        grp.launch()
        # + context term, LeakyReLU, SharedDropout of the MLPs (after the activation, nn/common.py:47-51): [B,1,H] masks shared over a
        # sentence's positions for head_ff, one value per ROW for the 2-D inputs of the other three (nn/dropout.py:52-53)
        _C.check(lib.vlg_ff_mlp_act(_C.ptr(X), _C.ptr(cterm), _C.ptr(drop_head), _C.ptr(drop_small), B, L, Ms, H, adt, SLOPE, st), "ff_mlp_act")
        W_nh, b_nh = Weff[0:2].reshape(2 * H, H), beff[0:2].reshape(2 * H)
        W_lr, b_lr = Weff[2:4].reshape(2 * H, H), beff[2:4].reshape(2 * H)
        Wv, bv = c(P["ff.mid_ff.valence_linear.weight"]), c(P["ff.mid_ff.valence_linear.bias"])
        Wd, bd = c(P["ff.mid_ff.direction_linear.weight"]), c(P["ff.mid_ff.direction_linear.bias"])
        W1_, b1_ = c(P["ff.mid_ff.linear1.weight"]), c(P["ff.mid_ff.linear1.bias"])
        A3 = torch.empty((M, 2, 2, H), dtype=act, device=dev)                           # [m,dir,val,c]
        wT = None
        if _fused(act, H):
            # every stage ONE launch: the rows stream through a product whose weight block sits in registers, and the element-wise pass runs on the
            # accumulators (vlg_ff_linear_act) -- no pre-activation tensor is written or read.  The backward pass's transposed weights in one launch.
            A1 = torch.empty((M, 2 * H), dtype=act, device=dev)
            A2, A4, A5 = (torch.empty((n, H), dtype=act, device=dev) for n in (2 * M, 4 * M, 4 * M))
            W_nh, b_nh, W_lr, b_lr, Wv, Wd, W1_ = (t.contiguous() for t in (W_nh, b_nh, W_lr, b_lr, Wv, Wd, W1_))
            _linear_act(X, W_nh, b_nh, A1, nb=2, residual=X, om=2, oy=1)                   # valence stage, nn/dmv_spec.py:41-44: act(bottleneck + x)
            _linear_act(A1.view(2 * M, H), Wv, bv, A2)                                     # h [M,val,H]
            _linear_act(A2, W_lr, b_lr, A3, nb=2, residual=X, rs=1, om=4, oy=2)            # direction stage, :46-50: rows (m,val) -> [m,dir,val]
            # direction_linear + nn.Dropout (:52) and the output stage (:52-54, linear2 folded into the projections) in ONE launch: A4's rows go
            # from the first product's epilogue to the second product through LDS (both are still written: the adjoint reads them)
            _linear_act_chain2(A3.view(4 * M, H), _stage(Wd, A4, bias=bd, mask=mid_mask, mask_scale=mid_scale, rng=mid_rng, p=p_mid), _stage(W1_, A5, bias=b1_))
            # linear1 | direction | valence | the (left, right) blocks | the (no, has) blocks, transposed: what the backward launches read
            wT = _transpose256([W1_, Wd, Wv, W_lr[:H], W_lr[H:], W_nh[:H], W_nh[H:]], torch.empty((7, H, H), dtype=act, device=dev))
        else:
            # ---- valence stage, nn/dmv_spec.py:41-44 ----
            A1 = torch.addmm(b_nh, X, W_nh.t())                                             # [M,2H] = (no | has) bottleneck outputs
            _act(A1, A1, M, 2, H, residual=X)                                               # act(bottleneck + x) (the skip connection)
            A2 = torch.addmm(bv, A1.view(2 * M, H), Wv.t())
            _act(A2, A2, 2 * M, 1, H)                                                       # h [M,val,H]
            # ---- direction stage, :46-50 ----
            Z = torch.addmm(b_lr, A2, W_lr.t())                                             # [2M,2H]: rows (m,val), columns (dir,c)
            _act(Z, A3, M, 4, H, residual=X, swap=True)
            A4 = torch.addmm(bd, A3.view(4 * M, H), Wd.t())
            _act(A4, A4, 4 * M, 1, H, mask=mid_mask, mask_scale=mid_scale, rng=mid_rng, p=p_mid)   # nn.Dropout after the direction stage (nn/dmv_spec.py:52)
            # ---- output stage, :52-54 with linear2 folded into the projections ----
            A5 = torch.addmm(b1_, A4, W1_.t())
            _act(A5, A5, 4 * M, 1, H)
        big = torch.addmm(bp[:2 * r], A5[:4 * M0], Wp[:2 * r].t())                      # [4 M0, 2r]: attach.project1 | dec.project1
        small = small_matmul(A5[4 * M0:], Wp[2 * r:].t(), bias=bp[2 * r:])                # [4 Ms, 4r]: attach.p2 | root.p2 | root.p1 | dec.p2
        # the scorers' inputs as VIEWS of the two products (vlgae_amd.scorer takes rows a constant stride apart in place); they are outputs
        # of this Function, so autograd hands their cotangents straight to backward -- no slice nodes, no copies
        x1 = big[:, :r].view(B, L, 2, 2, r)
        y1 = big[:, r:].view(B, L, 2, 2, r)
        x2 = small[:4 * T, :r].view(T, 2, 2, r)
        y2 = small[4 * T + 4:, 3 * r:].view(2, 2, 2, r)
        root_rule = torch.empty((T,), dtype=torch.float32, device=dev)                   # ldndmv.py:205: sum over (dir, val), softmax over tokens: one launch
        _C.check(lib.vlg_ff_root_rule(_C.ptr(small), 4 * r, T, r, adt, _C.ptr(root_rule), st), "ff_root_rule")
        ctx.save_for_backward(emb2, cmean, X, A1, A2, A3, A4, A5, We, Wc, W_nh, W_lr, Wv, Wd, W1_, W2_, b2_, PW, Wp, W0s, b0s, W1s,
                              *(c(t.detach()) for t in small_in), *(c(P[f"ff.{m}.linear.weight"]) for m in ("child_ff", "root_ff", "dec_ff")),
                              small, root_rule)
        ctx.wT = wT     # (an intermediate of this Function, not an input or output: kept on the context)
        ctx.drops = drops
        ctx.meta = (nb, B, L, E, h, T, H, r, act, [t.dtype for t in (emb, x, token_emb, root_emb, dec_emb)], [p.dtype for p in params])
        return x1, x2, y1, y2, root_rule

    @staticmethod
    @once_differentiable
    def backward(ctx, g_x1, g_x2, g_y1, g_y2, g_root):
        (emb2, cmean, X, A1, A2, A3, A4, A5, We, Wc, W_nh, W_lr, Wv, Wd, W1_, W2_, b2_, PW, Wp, W0s, b0s, W1s, tok, rootE, decE,
         Wchild, Wroot, Wdec, small, root_rule) = ctx.saved_tensors
        nb, B, L, E, h, T, H, r, act, in_dt, p_dt = ctx.meta
        M0, Ms = B * L, T + 3
        M = M0 + Ms
        dev = A5.device
        drop_mid = ctx.drops[2]
        mid_rng, mid_mask, p_mid = (drop_mid[0], None, drop_mid[1]) if isinstance(drop_mid, tuple) else (None, drop_mid, 0.0)
        lib, st, adt = _C.lib(), _C.stream_of(A5), _C.BF16 if act == torch.bfloat16 else _C.F32
        G = {}
        if (g_x1.dtype == act and g_y1.dtype == act and g_x1.stride() == g_y1.stride() == (L * 8 * r, 8 * r, 4 * r, 2 * r, 1)
                and g_y1.data_ptr() == g_x1.data_ptr() + r * g_x1.element_size()):
            g_big = g_x1.as_strided((4 * M0, 2 * r), (2 * r, 1), g_x1.storage_offset())   # the scorer's adjoint wrote them side by side
        else:
            g_big = torch.cat([g_x1.reshape(4 * M0, r), g_y1.reshape(4 * M0, r)], 1).to(act)
        # the cotangent of the small product in ONE pass: the scorers' g_x2 / g_y2 blocks, the root rule's adjoint (log-softmax backward,
        # d r2[c,dv,:] = dlogit[c] r1[dv,:], d r1[dv,:] = sum_c dlogit[c] r2[c,dv,:]), zeros elsewhere
        def rows_of(t, n):
            t = t.reshape(n, r)
            t = t if t.dtype == act else t.to(act)
            return t if t.stride(1) == 1 else t.contiguous()
        gx2, gy2 = rows_of(g_x2, 4 * T), rows_of(g_y2, 8)
        g_small = torch.empty((4 * Ms, 4 * r), dtype=act, device=dev)
        _C.check(lib.vlg_ff_root_rule_backward(_C.ptr(small), 4 * r, T, r, adt, _C.ptr(root_rule), _C.ptr(g_root.float().contiguous()), _C.ptr(gx2),
                                               gx2.stride(0), _C.ptr(gy2), gy2.stride(0), _C.ptr(g_small), st), "ff_root_rule_backward")
        # ---- folded projections ----
        gA5 = torch.empty_like(A5)
        wT = ctx.wT          # [7,H,H]: linear1 | direction | valence | left | right | no | has weights transposed (the fused launches' operand) or None
        # the 2r = 32 columns of the parents' cotangent: product and linear1's LeakyReLU' in one launch (the weight rows are the operand as they lie)
        fuse5 = wT is not None and 2 * r == 32 and g_big.stride(1) == 1 and g_big.stride(0) % 8 == 0 and g_big.data_ptr() % 16 == 0 and M0 > 0
        if fuse5:
            _linear_act_bwd(g_big, Wp[:2 * r], A5[:4 * M0], gA5[:4 * M0], w_kn=True)
        else:
            torch.mm(g_big, Wp[:2 * r], out=gA5[:4 * M0])
        # (everything below in the activations' dtype: as fp32 GEMMs on one workgroup each the library takes 30-50 us for these products)
        dWp, dbp = torch.empty((6 * r, H), dtype=act, device=dev), torch.empty((6 * r,), dtype=act, device=dev)
        wg = WgradGroup(lazy=True)   # the seven split-K weight gradients of this pass: ONE grid per kernel image at the end, their reductions one launch
        _wgrad(g_big, A5[:4 * M0], out=(dWp[:2 * r], dbp[:2 * r]), defer=wg)            # [2r,H], [2r]: split-K, written in place
        grp = SmallMatmulGroup()
        grp.add(g_small, Wp[2 * r:], out=gA5[4 * M0:])
        grp.add(g_small.t(), A5[4 * M0:], out=dWp[2 * r:])                              # [4r,H]: 4 (T + 3) rows
        grp.add(_ones(4 * Ms, act, dev).t(), g_small, out=dbp[2 * r:].unsqueeze(0))     # column sums as a product with ones (no reduce launch)
        grp.launch()
        # ---- linear1, direction ----
        if fuse5:
            _act_bwd(gA5[4 * M0:], A5[4 * M0:], gA5[4 * M0:], 4 * Ms, 1, H)                # (the token / root / decision rows: 4 (T + 3))
            g = gA5
        else:
            g = _act_bwd(gA5, A5, gA5, 4 * M, 1, H)
        G["linear1.w"], G["linear1.b"] = _wgrad(g, A4, dtype=act, defer=wg)
        gX = torch.empty((M, H), dtype=torch.float32, device=g.device)                   # the skip connections' cotangent
        gZ = torch.empty((2 * M, 2 * H), dtype=act, device=g.device)                     # [m,val,dir,c]
        if wT is not None:   # linear1's and direction_linear's adjoints (each: the input-gradient product + the element-wise adjoint in front of it) in ONE launch
            g4 = torch.empty_like(g)
            _linear_act_chain2(g, _stage(wT[0], g4, act=A4, mask=mid_mask, mask_scale=ctx.drops[3], rng=mid_rng, p=p_mid),
                               _stage(wT[1], gZ, act=A3, J=4, total=gX, swap=True), backward=True)   # [m,dir,val,c] -> [m,val,dir,c], gX = the four rows' sum
            g = g4
        else:
            g = g @ W1_
            _act_bwd(g, A4, g, 4 * M, 1, H, mask=mid_mask, mask_scale=ctx.drops[3], rng=mid_rng, p=p_mid)
        G["direction.w"], G["direction.b"] = _wgrad(g, A3.view(4 * M, H), dtype=act, defer=wg)
        if wT is None:
            g = g @ Wd                                                                   # [m,dir,val,c]
            _act_bwd(g, A3, gZ, M, 4, H, total=gX, swap=True)
        # the gradients of the folded weights straight into their stack, in the activations' dtype (what the unfold products read)
        dWeff, dbeff = torch.empty((4, H, H), dtype=act, device=g.device), torch.empty((4, H), dtype=act, device=g.device)   # no, has, left, right
        _wgrad(gZ, A2, out=(dWeff[2:4].view(2 * H, H), dbeff[2:4].view(2 * H)), defer=wg)
        # ---- valence ----
        if wT is not None:
            g = _linear_act_bwd(gZ, wT[3:5], A2, torch.empty((2 * M, H), dtype=act, device=gZ.device))   # 512 cotangent columns (dir, c)
        else:
            g = gZ @ W_lr
            _act_bwd(g, A2, g, 2 * M, 1, H)
        G["valence.w"], G["valence.b"] = _wgrad(g, A1.view(2 * M, H), dtype=act, defer=wg)
        if wT is not None:
            gY = _linear_act_bwd(g, wT[2], A1, torch.empty_like(g), J=2, total=gX, accumulate=True)
        else:
            gY = g @ Wv                                                                  # [M,2,H]
            _act_bwd(gY, A1, gY, M, 2, H, total=gX, accumulate=True)
        gY = gY.view(M, 2 * H)
        _wgrad(gY, X, out=(dWeff[0:2].view(2 * H, H), dbeff[0:2].view(2 * H)), defer=wg)
        # ---- MLPs: gpre = LeakyReLU'(X) * SharedDropout mask * (gX + gY W_nh) ----
        gpre = torch.empty((M, H), dtype=act, device=g.device)
        if wT is not None:   # gY W_nh (512 columns (no | has, c)) + the skip connections' sum, the MLPs' LeakyReLU' and SharedDropout masks in one launch
            _C.check(_C.lib().vlg_ff_linear_mlp_act_backward(_C.ptr(gY), gY.stride(0), _C.ptr(wT[5:7]), M, _C.ptr(gX), _C.ptr(X), _C.ptr(ctx.drops[0]),
                                                             _C.ptr(ctx.drops[1]), M0, L, _C.ptr(gpre), SLOPE, _C.stream_of(gpre)), "ff_linear_mlp_act_backward")
        else:
            gT = gY @ W_nh
            _C.check(_C.lib().vlg_ff_mlp_act_backward(_C.ptr(gX), _C.ptr(gT), _C.ptr(X), _C.ptr(ctx.drops[0]), _C.ptr(ctx.drops[1]),
                                                      _C.ptr(gpre), B, L, Ms, H, _adt(gpre), SLOPE, _C.stream_of(gpre)), "ff_mlp_act_backward")
        gb = gpre[:M0]
        dWh = torch.empty((H, E + h), dtype=act, device=dev)                             # head_ff's [H, E + h] gradient: both column blocks written in place
        dbh = torch.empty((H,), dtype=act, device=dev)
        _wgrad(gb, emb2, out=(dWh[:, :E], dbh), defer=wg)                                # [H,E], [H]
        wg.flush()
        g_emb = linear_kn(gb, We) if wT is not None and linear_kn_ok(gb, We) else gb @ We   # [M0,E]
        gc = gb.view(B, L, H).sum(1)                                                     # [B,H] (fp32 accumulation inside the reduction)
        # ---- the remaining products in weight space, ONE grouped launch: the context columns, the token / root / decision MLPs (weight, bias
        # as a product with ones, input gradients), the unfolding of the bottleneck pairs Weff = W1 W0, beff = W1 b0 + b1 ----
        grp = SmallMatmulGroup()
        dPW = grp.add(dWp, W2_.t(), rank1=(dbp, b2_))                                   # Wp = PW W2, bp = PW b2 + Pb: dPW = dWp W2^T + dbp b2^T
        G["linear2.w"], G["linear2.b"] = grp.add(PW.t(), dWp), grp.add(PW.t(), dbp.unsqueeze(-1)).squeeze(-1)   # P^T dWp, P^T dbp
        grp.add(gc.t(), cmean, out=dWh[:, E:])                                           # [H,h]
        g_cmean = grp.add(gc, Wc, alpha=1.0 / L)                                         # [B,h]: d mean_l
        G["head.w"], G["head.b"] = dWh, dbh
        o = M0
        g_small_in = []
        for name, inp, W in (("child", tok, Wchild), ("root", rootE, Wroot), ("dec", decE, Wdec)):
            n = inp.shape[0]
            gs = gpre[o:o + n]
            G[name + ".w"], G[name + ".b"] = grp.add(gs.t(), inp), grp.add(_ones(n, act, dev).t(), gs).squeeze(0)
            g_small_in.append(grp.add(gs, W))
            o += n
        if nb:   # (in the activations' dtype: the library's batched fp32 kernels take ~50 us each for these 40-MFLOP products)
            dW1s = grp.add(dWeff, W0s.transpose(1, 2), rank1=(dbeff, b0s))               # [4,H,nb] = dWeff W0^T + dbeff b0^T
            dW0s = grp.add(W1s.transpose(1, 2), dWeff)                                   # [4,nb,H] = W1^T dWeff
            db0s = grp.add(W1s.transpose(1, 2), dbeff.unsqueeze(-1)).squeeze(-1)         # [4,nb] = W1^T dbeff
        grp.launch()
        # ---- gradients in the order of param_names ----
        out = [G["head.w"], G["head.b"], G["child.w"], G["child.b"], G["root.w"], G["root.b"], G["dec.w"], G["dec.b"]]
        for k in range(4):
            out += [dW0s[k], db0s[k], dW1s[k], dbeff[k]] if nb else [dWeff[k], dbeff[k]]
        out += [G["valence.w"], G["valence.b"], G["direction.w"], G["direction.b"], G["linear1.w"], G["linear1.b"], G["linear2.w"],
                G["linear2.b"]]
        for k in range(6):
            out += [dPW[k * r:(k + 1) * r], dbp[k * r:(k + 1) * r]]
        # one multi-tensor conversion for the ~45 fp32 gradients instead of a cast launch each
        todo = [k for k, (t, d) in enumerate(zip(out, p_dt)) if t.dtype != d]
        if todo and len({p_dt[k] for k in todo}) == 1:
            flat = torch.empty(sum(out[k].numel() for k in todo), dtype=p_dt[todo[0]], device=out[todo[0]].device)
            dsts = [v.view(out[k].shape) for v, k in zip(flat.split_with_sizes([out[k].numel() for k in todo]), todo)]
            torch._foreach_copy_(dsts, [out[k] for k in todo])
            for k, v in zip(todo, dsts):
                out[k] = v
        else:
            out = [t if t.dtype == d else t.to(d) for t, d in zip(out, p_dt)]
        # d mean_l(x): every position of a sentence gets the same row -- cast the [B,h] rows, then broadcast as a stride-0 view (the
        # attention fuse's adjoint reads it in place; materialised it would be a [B,L,h] fp32 tensor)
        g_x = (g_cmean if g_cmean.dtype == in_dt[1] else g_cmean.to(in_dt[1])).unsqueeze(1).expand(B, L, h)
        ins = [g_emb.view(B, L, E), g_x, g_small_in[0], g_small_in[1], g_small_in[2]]
        ins = [t if t.dtype == d else t.to(d) for t, d in zip(ins, in_dt)]
        need = ctx.needs_input_grad
        return (None, None, *(t if n else None for t, n in zip(ins, need[2:7])), *(t if n else None for t, n in zip(out, need[7:])))


def parser_feed_forward(P, emb, x, token_emb=None, root_emb=None, dec_emb=None, drop_head=None, drop_small=None, drop_mid=None, mid_scale=1.0,
                        mid_rng=None, p_mid=0.0):
    """ldndmv.py:174-205 up to the scorers' projected inputs -> (x1 [B,L,2,2,r], x2 [T,2,2,r], y1 [B,L,2,2,r], y2 [2,2,2,r],
    root_rule [T]), the arguments of `scorer.ndmv_potentials`.

    P: dict of the reference modules' parameters under their own names behind "ff." ("ff.head_ff.linear.weight", ...,
    "ff.mid_ff.HASCHILD_linear.0.weight" (n_bottleneck > 0) or "ff.mid_ff.HASCHILD_linear.weight", ..., "ff.root_scorer.project2.bias");
    token_emb / root_emb / dec_emb default to P's entries of those names.  emb [B,L,E] (`encoded['emb']`), x [B,L,h] (the encodings the
    parser sees: the attention-fused copy, joint.py:670-675).
    Training-mode dropout as explicit masks (entries 0 or 1/(1-p); `dropout_masks` draws them): drop_head [B,H] -- head_ff's SharedDropout,
    one mask per sentence; drop_small [T+3] -- child_ff / root_ff / dec_ff see 2-D inputs, where SharedDropout's mask is one scalar per
    row (nn/dropout.py:52-53); drop_mid [4 (B L + T + 3), H] -- mid_ff's nn.Dropout after its direction stage (nn/dmv_spec.py:52), rows in
    the order (input row, direction, valence) with the B L parent rows first; mid_scale multiplies it (a 0 / 1 keep-mask with mid_scale = 1 / (1 - p)
    is nn.Dropout without the division pass over the 21 MB mask: `dropout_masks(..., scaled_mid=False)`).  None = identity (eval).
    mid_rng (an encoders.DeviceRng) with p_mid > 0 instead of drop_mid: mid_ff's keep-mask is drawn INSIDE the activation kernel of the
    direction stage and regenerated by its adjoint (counter-based, site SITE_MID_FF) -- no mask tensor at all."""
    _C.require_gpu(emb, "parser_feed_forward")
    token_emb = P["token_emb"] if token_emb is None else token_emb
    root_emb = P["root_emb"] if root_emb is None else root_emb
    dec_emb = P["dec_emb"] if dec_emb is None else dec_emb
    nb = P["ff.mid_ff.HASCHILD_linear.0.weight"].shape[0] if "ff.mid_ff.HASCHILD_linear.0.weight" in P else 0
    if emb.dtype not in (torch.float32, torch.bfloat16):
        raise ValueError(f"parser_feed_forward: float32 or bfloat16 embeddings, got {emb.dtype}")
    B, L, _ = emb.shape
    T = token_emb.shape[0]
    H = P["ff.head_ff.linear.weight"].shape[0]
    if tuple(x.shape[:2]) != (B, L) or tuple(root_emb.shape[:1]) != (1,) or tuple(dec_emb.shape[:1]) != (2,):
        raise ValueError(f"parser_feed_forward: emb {tuple(emb.shape)} x {tuple(x.shape)} root_emb {tuple(root_emb.shape)} dec_emb {tuple(dec_emb.shape)}")
    for name, m, shape in (("drop_head", drop_head, (B, H)), ("drop_small", drop_small, (T + 3,)), ("drop_mid", drop_mid, (4 * (B * L + T + 3), H))):
        if m is not None and tuple(m.shape) != shape:
            raise ValueError(f"parser_feed_forward: {name} must be {shape}, got {tuple(m.shape)}")
    if H % 8:
        raise ValueError(f"parser_feed_forward: hidden size {H} must be a multiple of 8")
    if mid_rng is not None and p_mid > 0:
        if drop_mid is not None:
            raise ValueError("parser_feed_forward: drop_mid and mid_rng exclude each other")
        mid = (mid_rng, float(p_mid))
    else:
        mid = None if drop_mid is None else drop_mid.detach().to(emb.dtype).contiguous()
    drops = (_mask32(drop_head), _mask32(drop_small), mid, float(mid_scale))
    return _ParserFF.apply(nb, drops, emb, x, token_emb, root_emb, dec_emb, *(P[k] for k in param_names(nb)))


def dropout_masks(B, L, T, H, p_ff=0.33, p_mid=0.3, device=None, dtype=torch.float32, generator=None, scaled_mid=True):
    """One training step's masks for `parser_feed_forward` at the shipped rates (config/model/vlgae.yaml: _dropout 0.33, mid_ff 0.3):
    (drop_head [B,H] float32, drop_small [T+3] float32, drop_mid [4 (B L + T + 3), H] in `dtype`, the activations'); a rate of 0 gives
    None.  The two small ones come out of ONE draw.  scaled_mid=False: drop_mid is the 0 / 1 keep-mask and a fourth value, 1 / (1 - p_mid),
    is returned for `mid_scale` (one launch less per step over the largest mask)."""
    def draw(shape, p, dt, scale=True):
        if p <= 0:
            return None
        m = torch.empty(shape, dtype=dt, device=device).bernoulli_(1 - p, generator=generator)
        return m.div_(1 - p) if scale else m
    small = draw((B * H + T + 3,), p_ff, torch.float32)
    head, rows = (None, None) if small is None else (small[:B * H].view(B, H), small[B * H:])
    mid = draw((4 * (B * L + T + 3), H), p_mid, dtype, scaled_mid)
    return (head, rows, mid) if scaled_mid else (head, rows, mid, 1.0 if mid is None else 1.0 / (1 - p_mid))
