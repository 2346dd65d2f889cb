"""Language-side factor features -- host-side mirror of `DependencyBoxRel.lang_feat_max_tree` (src/model/joint.py:235-292).

    lang_feat_max_tree(x, lengths, merged_dec, merged_attach, params)  ->  txt [B,2N,d], txt_mask [B,2N], txt_marginal [B,2N]

The reference runs ~25 torch ops here (DMV1o partition + autograd.grad, argmax + nonzero + scatter, gather, cat, masked mean,
three MLP encoders, gather, einsum, matmul, adds, cat).  Here the same values come from:
  DMV1o marginals ‖ Viterbi pass (two streams)            vlgae_amd.torch_struct (joint.py:251-258)
  one marginal / mask kernel                               vlg_langfeat_marginal   (:246-249, 258-262)
  one root-mean + cat kernel                               vlg_langfeat_root_cat   (:263-266)
  ONE projection GEMM for the word | child | parent encoders of the same x (library GEMM, concatenated weights) and one
  epilogue kernel (LeakyReLU, parent rows gathered by head, word rows straight into txt)     vlg_langfeat_split (:267-273)
  trilinear on the matrix cores + affine (library GEMM) + one epilogue into txt's arc half    vlg_trilinear, vlg_langfeat_arc_out (:278-288)
and a hand-written backward in the same granularity, with the encoders' weight / bias gradients on the split-K kernel
(vlg_linear_wgrad).  Activations are stored in the features' dtype: bf16 (BASELINE.json configs) with fp32 accumulation, or
float32 end to end (the reference's `precision: 32`); nothing is down-cast unasked.
"""
import torch
from torch.autograd.function import once_differentiable

from . import _C
from .align import _plain, linear_wgrad, _wgrad_ok


def txt_marginal_and_mask(grad_attach, heads, lengths, add_marginal=True, B=None, N=None):
    """joint.py:246-262 from the DP outputs: (txt_marginal [B,2N] float32, txt_mask [B,2N] bool), one launch.
    add_marginal=False needs no DP output (grad_attach / heads may be None with B, N given): cat([mask, mask]), :262."""
    if heads is not None:
        B, N = heads.shape
    dev = lengths.device
    marg = torch.empty((B, 2 * N), dtype=torch.float32, device=dev)
    mask = torch.empty((B, 2 * N), dtype=torch.bool, device=dev)
    _C.require_gpu(lengths, "txt_marginal_and_mask")
    _C.check(_C.lib().vlg_langfeat_marginal(_C.ptr(grad_attach), _C.ptr(heads), _C.ptr(lengths), B, N, int(bool(add_marginal)),
                                            _C.ptr(marg), _C.ptr(mask), _C.stream_of(lengths)), "langfeat_marginal")
    return marg, mask


def _act(x, compute_dtype):
    """Storage type of the stage's activations: the features' own dtype unless the caller asks otherwise.  float32 features are
    computed in float32 (the reference's `precision: 32`, config/trainer/train.yaml:20) -- never down-cast silently."""
    act = x.dtype if compute_dtype is None else compute_dtype
    if act not in (torch.float32, torch.bfloat16):
        raise ValueError(f"lang_feat: activations must be float32 or bfloat16, got {act}")
    return act


def shared_dropout_masks(B, d, p, n=3, device=None, generator=None):
    """`SharedDropout.get_mask` (nn/dropout.py:61-63) for n encoders at once: [B,n,d] float32 with entries 0 or 1/(1-p), one mask
    per sentence and encoder, shared over the positions.  p = 0 -> None (identity)."""
    if p <= 0:
        return None
    if not 0 < p < 1:
        raise ValueError(f"dropout probability {p}")
    return torch.empty((B, n, d), dtype=torch.float32, device=device).bernoulli_(1 - p, generator=generator).div_(1 - p)


def _check_drop(drop, B, n, d, dev):
    """A view into a wider draw is taken in place (rows contiguous, masks of a sentence adjacent, 16-byte aligned)."""
    if drop is None:
        return None
    if tuple(drop.shape) != (B, n, d) or drop.dtype != torch.float32 or drop.device != dev:
        raise ValueError(f"dropout masks must be float32 [B,{n},d]={(B, n, d)} on {dev}, got {drop.dtype} {tuple(drop.shape)} on {drop.device}")
    if drop.stride(2) != 1 or (n > 1 and drop.stride(1) != d) or drop.stride(0) % 4 or drop.stride(0) < n * d or drop.data_ptr() % 16:
        drop = drop.contiguous()
    return drop


def _wgrad_into(dy, x, d_w, d_b, x_colsum=False):
    """d_w, d_b <- (dy^T x, sum_rows dy) [or, x_colsum: sum_rows x]: the split-K kernel for bf16, the library in fp32."""
    if dy.dtype == x.dtype and _wgrad_ok(x.shape[0], dy.shape[1], x.shape[1], dy.dtype):
        linear_wgrad(dy, x, want_x_colsum=x_colsum, out=(d_w, d_b))
    elif d_w.dtype == torch.float32:
        torch.matmul(dy.float().t(), x.float(), out=d_w)
        torch.sum((x if x_colsum else dy).float(), 0, out=d_b)
    else:
        d_w.copy_(dy.float().t() @ x.float())
        d_b.copy_((x if x_colsum else dy).float().sum(0))


class _LangFeat(torch.autograd.Function):
    """x [B,L,h], heads [B,N] -> txt [B,2N,d] (word_repr | arc_repr) in the activations' dtype, differentiable in x and every
    parameter."""

    @staticmethod
    def forward(ctx, x, lengths, heads, w_enc, b_enc, w1, w2, b_arc, slope, aux, drop, act, pre_given):
        """pre_given: the three encoders' shared projection [B,N,3d] of `encoder_projection` (then x / w_enc / b_enc are not read and get
        no gradient here: the cotangent of the projection is returned instead) or None (the projection is computed here)."""
        if pre_given is not None:
            B, N, d3 = pre_given.shape
            L, h, d = N - 1, 0, d3 // 3
            dev = pre_given.device
            st = _C.stream_of(pre_given)
        else:
            B, L, h = x.shape
            N, d = L + 1, w_enc.shape[0] // 3
            dev = x.device
            st = _C.stream_of(x)
        M, lib = B * N, _C.lib()
        adt = _C.BF16 if act == torch.bfloat16 else _C.F32
        if pre_given is not None:
            pre, x1, w_enc_c = pre_given.detach().reshape(M, 3 * d), None, None
        else:
            dt, x_c = _C.in_dtype(x.detach())
            x1 = torch.empty((M, h), dtype=act, device=dev)
            _C.check(lib.vlg_langfeat_root_cat(_C.ptr(x_c), _C.ptr(lengths), B, L, h, dt, _C.ptr(x1), adt, st), "langfeat_root_cat")
            w_enc_c = w_enc.detach().to(act)
            pre = torch.addmm(b_enc.detach().to(act), x1, w_enc_c.t())                  # [M,3d]: the three encoders' Linear
        if not torch.is_tensor(heads):   # a StructureHandle: the DPs ran on side streams beside the two launches above
            heads = heads.wait()[2]
        txt = torch.empty((B, 2 * N, d), dtype=act, device=dev)
        child, parent, cps = (torch.empty((M, d), dtype=act, device=dev) for _ in range(3))
        _C.check(lib.vlg_langfeat_split(_C.ptr(pre), _C.ptr(heads), _C.ptr(drop), 0 if drop is None else drop.stride(0), B, N, d, adt, float(slope), _C.ptr(txt), _C.ptr(child),
                                        _C.ptr(parent), _C.ptr(cps), st), "langfeat_split")
        w1_c = w1.detach().to(act).contiguous()
        tbytes = lib.vlg_trilinear_workspace(M, d, d, d, adt)
        (tri,), tws = _C.alloc_f32(dev, ((M, d),), tbytes)
        _C.check(lib.vlg_trilinear_ws(_C.ptr(child), _C.ptr(w1_c), _C.ptr(parent), M, d, d, d, adt, _C.ptr(tws) if tbytes else None,
                                      tbytes, _C.ptr(tri), st), "trilinear")
        w2_c = w2.detach().to(act)
        aff = torch.addmm(b_arc.detach().to(act), cps, w2_c)                               # (child + parent) w2 + b
        _C.check(lib.vlg_langfeat_arc_out(_C.ptr(tri), _C.ptr(aff), B, N, d, adt, _C.ptr(txt), st), "langfeat_arc_out")
        if aux is not None:   # inspection (tests): the encoders' activations, whose signs are the LeakyReLU branches the adjoint takes
            aux.update(child=child.view(B, N, d), parent=parent.view(B, N, d))
        ctx.save_for_backward(x1, child, parent, cps, heads, lengths, w_enc_c, w1_c, w2_c, drop)
        ctx.shared = pre_given is not None
        ctx.meta = (B, L, h, d, float(slope), act, None if x is None else x.dtype, None if w_enc is None else w_enc.dtype, None if b_enc is None else b_enc.dtype,
                    w1.dtype, w2.dtype, b_arc.dtype)
        return txt

    @staticmethod
    @once_differentiable
    def backward(ctx, d_txt):
        x1, child, parent, cps, heads, lengths, w_enc_c, w1_c, w2_c, drop = ctx.saved_tensors
        B, L, h, d, slope, act, t_x, t_wenc, t_benc, t_w1, t_w2, t_barc = ctx.meta
        N, M = L + 1, B * (L + 1)
        dev, lib, st = child.device, _C.lib(), _C.stream_of(child)
        bf = torch.bfloat16
        adt = _C.BF16 if act == bf else _C.F32
        if d_txt.dtype not in (torch.float32, bf) or not d_txt.is_contiguous():
            d_txt = d_txt.to(act).contiguous()
        # every parameter gradient lives in one fp32 allocation: a single cast launch at the end
        nbytes = lib.vlg_trilinear_backward_workspace(M, d, d, d, adt)
        outs, ws = _C.alloc_f32(dev, ((3 * d, max(h, 1)), (3 * d,), (d, d, d), (d, d), (d,), (M, d), (M, d)), nbytes)
        d_wenc, d_benc, d_w1, d_w2, d_barc, d_child, d_parent = outs
        # ---- arc half: g = d arc_repr [M,d], one contiguous copy in the activations' dtype ----
        gb = d_txt[:, N:, :].to(act).reshape(M, d) if d_txt.dtype != act else d_txt[:, N:, :].reshape(M, d)
        if gb.stride(1) != 1 or gb.stride(0) != d:
            gb = gb.contiguous()
        _C.check(lib.vlg_trilinear_backward_g(_C.ptr(child), _C.ptr(w1_c), _C.ptr(parent), _C.ptr(gb), adt, M, d, d, d, adt,
                                              _C.ptr(ws), nbytes, _C.ptr(d_child), _C.ptr(d_w1), _C.ptr(d_parent), st), "trilinear_backward")
        d_sum = gb @ w2_c.t()                                                               # d (child + parent)
        _wgrad_into(cps, gb, d_w2, d_barc, x_colsum=True)                                   # w2 is stored [in, out]: cps^T g, sum_rows g
        # ---- encoders ----
        d_pre = torch.empty((M, 3 * d), dtype=act, device=dev)
        _C.check(lib.vlg_langfeat_split_backward(_C.ptr(d_txt), _C.BF16 if d_txt.dtype == bf else _C.F32, _C.ptr(d_child),
                                                 _C.ptr(d_parent), _C.ptr(d_sum), adt, _C.ptr(child), _C.ptr(parent), _C.ptr(heads),
                                                 _C.ptr(drop), 0 if drop is None else drop.stride(0), B, N, d, adt, slope, _C.ptr(d_pre), st),
                 "langfeat_split_backward")
        need = ctx.needs_input_grad
        if ctx.shared:   # the projection belongs to `encoder_projection`: hand its cotangent over (that Function does the encoders' adjoint once)
            pdt = (t_w1, t_w2, t_barc)
            if all(t == pdt[0] for t in pdt) and pdt[0] != torch.float32:
                pg = outs.cast(5, pdt[0])[2:5]                                              # one conversion launch for the three (the two unused slots are 4 d floats)
            else:
                pg = [g if g.dtype == t else g.to(t) for g, t in zip([d_w1, d_w2, d_barc], pdt)]
            return (None, None, None, None, None, *(g if n else None for g, n in zip(pg, need[5:8])), None, None, None, None, d_pre.view(B, N, 3 * d))
        _wgrad_into(d_pre, x1, d_wenc, d_benc)
        d_x1 = d_pre @ w_enc_c                                                              # [M,h], library GEMM
        xdt = t_x if t_x in (torch.float32, torch.bfloat16) else torch.float32
        d_x = torch.empty((B, L, h), dtype=xdt, device=dev)
        _C.check(lib.vlg_langfeat_root_cat_backward(_C.ptr(d_x1), _C.ptr(lengths), B, L, h, adt, _C.ptr(d_x),
                                                    _C.BF16 if xdt == torch.bfloat16 else _C.F32, st), "langfeat_root_cat_backward")
        pdt = (t_wenc, t_benc, t_w1, t_w2, t_barc)
        pg = [d_wenc, d_benc, d_w1, d_w2, d_barc]
        if all(t == pdt[0] for t in pdt) and pdt[0] != torch.float32:
            pg = outs.cast(5, pdt[0])
        else:
            pg = [g if g.dtype == t else g.to(t) for g, t in zip(pg, pdt)]
        return ((d_x if d_x.dtype == t_x else d_x.to(t_x)) if need[0] else None, None, None,
                *(g if n else None for g, n in zip(pg, need[3:8])), None, None, None, None, None)


class _EncProject(torch.autograd.Function):
    """x [B,L,h] -> pre [B,N,3d] = cat([masked mean, x]) W_cat^T + b_cat: the Linear layers of the word | child | parent encoders on the
    root-augmented encodings (joint.py:204-209 and :262-273 read the SAME x1 through the SAME word encoder), computed ONCE per step."""

    @staticmethod
    def forward(ctx, x, lengths, w_enc, b_enc, act):
        B, L, h = x.shape
        N, d3 = L + 1, w_enc.shape[0]
        M, dev, lib, st = B * N, x.device, _C.lib(), _C.stream_of(x)
        adt = _C.BF16 if act == torch.bfloat16 else _C.F32
        dt, x_c = _C.in_dtype(x.detach())
        x1 = torch.empty((M, h), dtype=act, device=dev)
        _C.check(lib.vlg_langfeat_root_cat(_C.ptr(x_c), _C.ptr(lengths), B, L, h, dt, _C.ptr(x1), adt, st), "langfeat_root_cat")
        w_c = w_enc.detach().to(act)
        pre = torch.addmm(b_enc.detach().to(act), x1, w_c.t())
        ctx.save_for_backward(x1, lengths, w_c)
        ctx.meta = (B, L, h, d3, act, x.dtype, w_enc.dtype, b_enc.dtype)
        return pre.view(B, N, d3)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x1, lengths, w_c = ctx.saved_tensors
        B, L, h, d3, act, t_x, t_w, t_b = ctx.meta
        N, M = L + 1, B * (L + 1)
        dev, lib, st = x1.device, _C.lib(), _C.stream_of(x1)
        adt = _C.BF16 if act == torch.bfloat16 else _C.F32
        g = g.to(act).contiguous().view(M, d3)
        pdt = t_w if t_w == t_b and t_w in (torch.float32, torch.bfloat16) else torch.float32
        d_w, d_b = torch.empty((d3, h), dtype=pdt, device=dev), torch.empty((d3,), dtype=pdt, device=dev)
        _wgrad_into(g, x1, d_w, d_b)
        d_x1 = g @ w_c
        xdt = t_x if t_x in (torch.float32, torch.bfloat16) else torch.float32
        d_x = torch.empty((B, L, h), dtype=xdt, device=dev)
        _C.check(lib.vlg_langfeat_root_cat_backward(_C.ptr(d_x1), _C.ptr(lengths), B, L, h, adt, _C.ptr(d_x),
                                                    _C.BF16 if xdt == torch.bfloat16 else _C.F32, st), "langfeat_root_cat_backward")
        need = ctx.needs_input_grad
        return ((d_x if d_x.dtype == t_x else d_x.to(t_x)) if need[0] else None, None, (d_w if pdt == t_w else d_w.to(t_w)) if need[2] else None,
                (d_b if pdt == t_b else d_b.to(t_b)) if need[3] else None, None)


class _WordFromPre(torch.autograd.Function):
    """pre [B,N,3d] -> word_repr [B,N,d] = word third * SharedDropout mask; the adjoint writes the full-width cotangent of pre (zeros for
    the child | parent thirds) in one pass, which autograd adds to lang_feat_max_tree's."""

    @staticmethod
    def forward(ctx, pre, drop):
        B, N, d3 = pre.shape
        d = d3 // 3
        p = pre.detach()
        adt = _C.BF16 if p.dtype == torch.bfloat16 else _C.F32
        out = torch.empty((B, N, d), dtype=p.dtype, device=p.device)
        _C.check(_C.lib().vlg_langfeat_rowscale(_C.ptr(p), d3, _C.ptr(drop), B, N, d, 0 if drop is None else drop.stride(0), adt, _C.ptr(out), d, d,
                                                _C.stream_of(p)), "langfeat_rowscale")
        ctx.drop, ctx.shape = drop, (B, N, d)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        B, N, d = ctx.shape
        drop = ctx.drop
        g = g.contiguous()
        adt = _C.BF16 if g.dtype == torch.bfloat16 else _C.F32
        d_pre = torch.empty((B, N, 3 * d), dtype=g.dtype, device=g.device)
        _C.check(_C.lib().vlg_langfeat_rowscale(_C.ptr(g), d, _C.ptr(drop), B, N, d, 0 if drop is None else drop.stride(0), adt, _C.ptr(d_pre), 3 * d, 3 * d,
                                                _C.stream_of(g)), "langfeat_rowscale")
        return d_pre, None


def encoder_projection(x, lengths, w_enc, b_enc, compute_dtype=None):
    """The Linear layers of the word | child | parent encoders on x1 = cat([masked mean of the words, x]) (joint.py:204-209, :262-273):
    pre [B,N,3d] in the features' dtype.  `lang_feat_word_only` and `lang_feat_max_tree` both start from this projection of the SAME
    un-fused encodings (the word encoder is evaluated twice by the reference, with two SharedDropout masks): pass the result as `pre=` to
    both and the root row, the GEMM and the encoders' adjoint (weight / bias / input gradients) run once per step instead of twice.
    w_enc [3d,h] / b_enc [3d]: the three encoders' `linear` parameters stacked word | child | parent."""
    x = _plain(x)
    _C.require_gpu(x, "encoder_projection")
    B, L, h = x.shape
    if w_enc.dim() != 2 or w_enc.shape[1] != h or w_enc.shape[0] % 24 or tuple(b_enc.shape) != (w_enc.shape[0],):
        raise ValueError(f"encoder_projection: w_enc {tuple(w_enc.shape)} b_enc {tuple(b_enc.shape)} for x {tuple(x.shape)} (3 d rows, d a multiple of 8)")
    if lengths.dtype != torch.int64:
        raise ValueError("encoder_projection: lengths must be int64 [B]")
    return _EncProject.apply(x, lengths.contiguous(), w_enc, b_enc, _act(x, compute_dtype))


class _WordOnly(torch.autograd.Function):
    """x [B,L,h] -> word_encoder(cat([masked mean, x])) [B,N,d] (joint.py:193-211), differentiable in x and the encoder."""

    @staticmethod
    def forward(ctx, x, lengths, w_word, b_word, drop, act):
        B, L, h = x.shape
        N, d = L + 1, w_word.shape[0]
        M, dev, lib, st = B * N, x.device, _C.lib(), _C.stream_of(x)
        adt = _C.BF16 if act == torch.bfloat16 else _C.F32
        dt, x_c = _C.in_dtype(x.detach())
        x1 = torch.empty((M, h), dtype=act, device=dev)
        _C.check(lib.vlg_langfeat_root_cat(_C.ptr(x_c), _C.ptr(lengths), B, L, h, dt, _C.ptr(x1), adt, st), "langfeat_root_cat")
        w_c = w_word.detach().to(act)
        out = torch.addmm(b_word.detach().to(act), x1, w_c.t())
        if drop is not None:
            _C.check(lib.vlg_langfeat_rowscale(_C.ptr(out), d, _C.ptr(drop), B, N, d, drop.stride(0), adt, _C.ptr(out), d, d, st), "langfeat_rowscale")
        ctx.save_for_backward(x1, lengths, w_c, drop)
        ctx.meta = (B, L, h, d, act, x.dtype, w_word.dtype, b_word.dtype)
        return out.view(B, N, d)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x1, lengths, w_c, drop = ctx.saved_tensors
        B, L, h, d, act, t_x, t_w, t_b = ctx.meta
        N, M = L + 1, B * (L + 1)
        dev, lib, st = x1.device, _C.lib(), _C.stream_of(x1)
        adt = _C.BF16 if act == torch.bfloat16 else _C.F32
        g = g.to(act).contiguous().view(M, d)
        if drop is not None:
            gs = torch.empty_like(g)
            _C.check(lib.vlg_langfeat_rowscale(_C.ptr(g), d, _C.ptr(drop), B, N, d, drop.stride(0), adt, _C.ptr(gs), d, d, st), "langfeat_rowscale")
            g = gs
        # gradients leave in the types of what receives them: no cast launches behind the kernels
        pdt = t_w if t_w == t_b and t_w in (torch.float32, torch.bfloat16) else torch.float32
        d_w, d_b = torch.empty((d, h), dtype=pdt, device=dev), torch.empty((d,), dtype=pdt, device=dev)
        _wgrad_into(g, x1, d_w, d_b)
        d_x1 = g @ w_c
        xdt = t_x if t_x in (torch.float32, torch.bfloat16) else torch.float32
        d_x = torch.empty((B, L, h), dtype=xdt, device=dev)
        _C.check(lib.vlg_langfeat_root_cat_backward(_C.ptr(d_x1), _C.ptr(lengths), B, L, h, adt, _C.ptr(d_x),
                                                    _C.BF16 if xdt == torch.bfloat16 else _C.F32, st), "langfeat_root_cat_backward")
        need = ctx.needs_input_grad
        return ((d_x if d_x.dtype == t_x else d_x.to(t_x)) if need[0] else None, None, (d_w if pdt == t_w else d_w.to(t_w)) if need[2] else None,
                (d_b if pdt == t_b else d_b.to(t_b)) if need[3] else None, None, None)


def lang_feat_word_only(x, lengths, w_word=None, b_word=None, drop=None, compute_dtype=None, pre=None, masks=True):
    """`DependencyBoxRel.lang_feat_word_only` (joint.py:193-211) -> (word_repr [B,N,d], mask [B,N] bool, mask as float32):
    root row = masked mean of the word encodings, then the word encoder (`MLP` without activation, config/model/vlgae.yaml:69-73).
    w_word [d,h] / b_word [d]: its Linear (nn.Linear layout); drop [B,1,d] float32 or None: its SharedDropout mask (training).
    pre = `encoder_projection(x, ...)` [B,N,3d]: the shared projection (x / w_word / b_word are then not read).
    masks=False: (word_repr, None, None) -- the caller only fuses the word features (joint.py:667-674 does not read the mask): one launch less."""
    if lengths.dtype != torch.int64:
        raise ValueError("lang_feat_word_only: lengths must be int64 [B]")
    lengths = lengths.contiguous()
    if pre is not None:
        _C.require_gpu(pre, "lang_feat_word_only")
        B, N, d3 = pre.shape
        L, d = N - 1, d3 // 3
        drop = _check_drop(drop, B, 1, d, pre.device)
        word = _WordFromPre.apply(pre, drop)
    else:
        x = _plain(x)
        _C.require_gpu(x, "lang_feat_word_only")
        B, L, h = x.shape
        d = w_word.shape[0]
        if tuple(w_word.shape) != (d, h) or tuple(b_word.shape) != (d,) or d % 8:
            raise ValueError(f"lang_feat_word_only: w_word {tuple(w_word.shape)} b_word {tuple(b_word.shape)} for x {tuple(x.shape)} (d a multiple of 8)")
        act = _act(x, compute_dtype)
        drop = _check_drop(drop, B, 1, d, x.device)
        word = _WordOnly.apply(x, lengths, w_word, b_word, drop, act)
    if not masks:
        return word, None, None
    with torch.no_grad():
        marg, mask = txt_marginal_and_mask(None, None, lengths, add_marginal=False, B=B, N=L + 1)
    return word, mask[:, :L + 1], marg[:, :L + 1]


def arc_word_features(x, lengths, heads, w_enc, b_enc, w1, w2, b_arc, slope=0.01, aux=None, drop=None, compute_dtype=None, pre=None):
    """joint.py:262-288: txt = cat([word_encoder(x1), arc_repr]) with x1 = cat([masked mean, x]) -- [B,2N,d] in the features'
    dtype (float32 features are computed in float32; `compute_dtype=torch.bfloat16` asks for bf16 storage explicitly).

    x [B,L,h]; lengths [B] int64; heads [B,N] int64 (`predicted`, joint.py:256-258) or the handle of `start_structure` (joined
    inside, after the launches that do not need the heads);
    w_enc [3d,h] / b_enc [3d]: the word | child | parent encoders' Linear parameters concatenated along the output
    dimension (nn.Linear layout [out,in]; word: no activation, child / parent: LeakyReLU(slope) -- config/model/vlgae.yaml:69-73,
    joint.py:216-222); w1 [d,d,d], w2 [d,d], b_arc [d]: the arc encoder (joint.py:223-232).
    drop [B,3,d] float32 or None: the three encoders' SharedDropout masks of this step (`shared_dropout_masks`; training mode,
    p = 0.33 in the shipped config); None = identity (eval / p = 0).
    aux: optional dict that receives the child / parent activations [B,N,d] (inspection only).
    pre = `encoder_projection(x, lengths, w_enc, b_enc)` [B,N,3d]: the shared projection (x / w_enc / b_enc are then not read here)."""
    d = w1.shape[0]
    if pre is not None:
        _C.require_gpu(pre, "arc_word_features")
        B, L = pre.shape[0], pre.shape[1] - 1
        if pre.shape[2] != 3 * d or tuple(w1.shape) != (d, d, d) or tuple(w2.shape) != (d, d):
            raise ValueError(f"arc_word_features: pre {tuple(pre.shape)} w1 {tuple(w1.shape)} w2 {tuple(w2.shape)}")
    else:
        x = _plain(x)
        _C.require_gpu(x, "arc_word_features")
        B, L, h = x.shape
        if tuple(w_enc.shape) != (3 * d, h) or tuple(b_enc.shape) != (3 * d,) or tuple(w1.shape) != (d, d, d) or tuple(w2.shape) != (d, d):
            raise ValueError(f"arc_word_features: w_enc {tuple(w_enc.shape)} b_enc {tuple(b_enc.shape)} w1 {tuple(w1.shape)} w2 {tuple(w2.shape)} "
                             f"for x {tuple(x.shape)}")
    if torch.is_tensor(heads):
        if tuple(heads.shape) != (B, L + 1) or heads.dtype != torch.int64:
            raise ValueError("arc_word_features: heads must be int64 [B,L+1]")
        heads = heads.contiguous()
    if lengths.dtype != torch.int64:
        raise ValueError("arc_word_features: lengths must be int64 [B]")
    if d % 16 or d > 128 or d not in (32, 64, 128):
        raise ValueError(f"arc_word_features: matching width d={d} (supported: 32, 64, 128)")
    if pre is not None:
        drop = _check_drop(drop, B, 3, d, pre.device)
        return _LangFeat.apply(None, lengths.contiguous(), heads, None, None, w1, w2, b_arc, float(slope), aux, drop, _act(pre, compute_dtype), pre)
    act = _act(x, compute_dtype)
    drop = _check_drop(drop, B, 3, d, x.device)
    return _LangFeat.apply(x, lengths.contiguous(), heads, w_enc, b_enc, w1, w2, b_arc, float(slope), aux, drop, act, None)


def start_structure(merged_dec, merged_attach, lengths, keep_viterbi=False):
    """The two DPs of lang_feat_max_tree (joint.py:251-258) started on side streams as soon as the potentials exist -- before the
    attention-fuse that produces `x` -- so that they overlap it: pass the returned handle as `structure=` to lang_feat_max_tree."""
    import vlgae_amd.torch_struct as ts
    with torch.no_grad():
        return ts.DMV1o([merged_dec.detach(), merged_attach.detach()], lengths).marginals_and_heads_async(keep_viterbi)


def lang_feat_max_tree(x, lengths, merged_dec, merged_attach, w_enc, b_enc, w1, w2, b_arc, add_marginal=True, slope=0.01,
                       keep_viterbi=False, aux=None, structure=None, drop=None, compute_dtype=None, pre=None, heads=None):
    """`DependencyBoxRel.lang_feat_max_tree` (joint.py:235-292) -> (txt [B,2N,d] in x's dtype, txt_mask [B,2N] bool, txt_marginal
    [B,2N] float32).  drop [B,3,d]: the word | child | parent encoders' SharedDropout masks (training; `shared_dropout_masks`).  The potentials are constants of this stage (detached, joint.py:252-253).  `structure` = the handle of an
    earlier `start_structure(...)` (then merged_dec / merged_attach / keep_viterbi are not used here).  pre = `encoder_projection(...)`: the
    encoders' projection shared with `lang_feat_word_only` (x / w_enc / b_enc are then not read here).  heads [B,N] (optional): use this
    tree instead of the Viterbi tree of the potentials (teacher forcing: the parity tests hand over the reference's `predicted`, so that a
    bf16 run is compared on the SAME tree and rounding is not mistaken for a flipped attachment).
    Measured: the two DPs are joined BEFORE the root row and the projection GEMM.  Letting those launches run beside the DPs
    (they do not need the heads) made the training step 90 us SLOWER (1.39 -> 1.48 ms as one HIP graph, same box): each DP is one
    workgroup per CU on a 93 us critical path, and a workgroup that has to wait for a CU behind a GEMM tile lengthens that path."""
    import vlgae_amd.torch_struct as ts
    with torch.no_grad():
        if structure is not None:
            _, marg, heads = structure.wait()
        else:
            marg, own_heads = ts.DMV1o([merged_dec.detach(), merged_attach.detach()], lengths).marginals_and_heads(keep_viterbi)
            heads = own_heads if heads is None else heads.to(own_heads.device, own_heads.dtype)   # teacher forcing (tests): the caller's tree
        txt_marginal, txt_mask = txt_marginal_and_mask(marg, heads, lengths, add_marginal)
    if aux is not None:
        aux["heads"] = heads
    txt = arc_word_features(x, lengths, heads, w_enc, b_enc, w1, w2, b_arc, slope, aux, drop, compute_dtype, pre)
    return txt, txt_mask, txt_marginal
