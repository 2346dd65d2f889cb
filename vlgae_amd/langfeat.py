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
(vlg_linear_wgrad).  bf16 activations, fp32 accumulation (BASELINE.json configs: bf16).
"""
import torch
from torch.autograd.function import once_differentiable

from . import _C
from .align import _plain, linear_wgrad, _wgrad_ok


def txt_marginal_and_mask(grad_attach, heads, lengths, add_marginal=True):
    """joint.py:246-262 from the DP outputs: (txt_marginal [B,2N] float32, txt_mask [B,2N] bool), one launch."""
    B, N = heads.shape
    dev = heads.device
    marg = torch.empty((B, 2 * N), dtype=torch.float32, device=dev)
    mask = torch.empty((B, 2 * N), dtype=torch.bool, device=dev)
    _C.check(_C.lib().vlg_langfeat_marginal(_C.ptr(grad_attach), _C.ptr(heads), _C.ptr(lengths), B, N, int(bool(add_marginal)),
                                            _C.ptr(marg), _C.ptr(mask), _C.stream_of(heads)), "langfeat_marginal")
    return marg, mask


class _LangFeat(torch.autograd.Function):
    """x [B,L,h], heads [B,N] -> txt [B,2N,d] bf16 (word_repr | arc_repr), differentiable in x and every parameter."""

    @staticmethod
    def forward(ctx, x, lengths, heads, w_enc, b_enc, w1, w2, b_arc, slope, aux):
        B, L, h = x.shape
        N, d = L + 1, w_enc.shape[0] // 3
        M, dev, lib = B * N, x.device, _C.lib()
        st = _C.stream_of(x)
        bf = torch.bfloat16
        dt, x_c = _C.in_dtype(x.detach())
        x1 = torch.empty((M, h), dtype=bf, device=dev)
        _C.check(lib.vlg_langfeat_root_cat(_C.ptr(x_c), _C.ptr(lengths), B, L, h, dt, _C.ptr(x1), st), "langfeat_root_cat")
        w_enc_c = w_enc.detach().to(bf)
        pre = torch.addmm(b_enc.detach().to(bf), x1, w_enc_c.t())                       # [M,3d]: the three encoders' Linear
        if not torch.is_tensor(heads):   # a StructureHandle: the DPs ran on side streams beside the two launches above
            heads = heads.wait()[2]
        txt = torch.empty((B, 2 * N, d), dtype=bf, device=dev)
        child, parent, cps = (torch.empty((M, d), dtype=bf, device=dev) for _ in range(3))
        _C.check(lib.vlg_langfeat_split(_C.ptr(pre), _C.ptr(heads), B, N, d, float(slope), _C.ptr(txt), _C.ptr(child),
                                        _C.ptr(parent), _C.ptr(cps), st), "langfeat_split")
        w1_c = w1.detach().to(bf).contiguous()
        tbytes = lib.vlg_trilinear_workspace(M, d, d, d, _C.BF16)
        (tri,), tws = _C.alloc_f32(dev, ((M, d),), tbytes)
        _C.check(lib.vlg_trilinear_ws(_C.ptr(child), _C.ptr(w1_c), _C.ptr(parent), M, d, d, d, _C.BF16, _C.ptr(tws) if tbytes else None,
                                      tbytes, _C.ptr(tri), st), "trilinear")
        w2_c = w2.detach().to(bf)
        aff = torch.addmm(b_arc.detach().to(bf), cps, w2_c)                                # (child + parent) w2 + b
        _C.check(lib.vlg_langfeat_arc_out(_C.ptr(tri), _C.ptr(aff), B, N, d, _C.ptr(txt), st), "langfeat_arc_out")
        if aux is not None:   # inspection (tests): the encoders' activations, whose signs are the LeakyReLU branches the adjoint takes
            aux.update(child=child.view(B, N, d), parent=parent.view(B, N, d))
        ctx.save_for_backward(x1, child, parent, cps, heads, lengths, w_enc_c, w1_c, w2_c)
        ctx.meta = (B, L, h, d, float(slope), x.dtype, w_enc.dtype, b_enc.dtype, w1.dtype, w2.dtype, b_arc.dtype)
        return txt

    @staticmethod
    @once_differentiable
    def backward(ctx, d_txt):
        x1, child, parent, cps, heads, lengths, w_enc_c, w1_c, w2_c = ctx.saved_tensors
        B, L, h, d, slope, t_x, t_wenc, t_benc, t_w1, t_w2, t_barc = ctx.meta
        N, M = L + 1, B * (L + 1)
        dev, lib, st = x1.device, _C.lib(), _C.stream_of(x1)
        bf = torch.bfloat16
        if d_txt.dtype not in (torch.float32, bf) or not d_txt.is_contiguous():
            d_txt = d_txt.to(bf).contiguous()
        # every parameter gradient lives in one fp32 allocation: a single cast launch at the end
        nbytes = lib.vlg_trilinear_backward_workspace(M, d, d, d, _C.BF16)
        outs, ws = _C.alloc_f32(dev, ((3 * d, h), (3 * d,), (d, d, d), (d, d), (d,), (M, d), (M, d)), nbytes)
        d_wenc, d_benc, d_w1, d_w2, d_barc, d_child, d_parent = outs
        # ---- arc half: g = d arc_repr [M,d] ----
        gb = d_txt[:, N:, :].to(bf).reshape(M, d) if d_txt.dtype != bf else d_txt[:, N:, :].reshape(M, d)   # one contiguous copy
        if gb.stride(1) != 1 or gb.stride(0) != d:
            gb = gb.contiguous()
        _C.check(lib.vlg_trilinear_backward_g(_C.ptr(child), _C.ptr(w1_c), _C.ptr(parent), _C.ptr(gb), _C.BF16, M, d, d, d, _C.BF16,
                                              _C.ptr(ws), nbytes, _C.ptr(d_child), _C.ptr(d_w1), _C.ptr(d_parent), st), "trilinear_backward")
        d_sum = gb @ w2_c.t()                                                               # d (child + parent), bf16
        if _wgrad_ok(M, d, d, bf):
            linear_wgrad(cps, gb, want_x_colsum=True, out=(d_w2, d_barc))                  # w2 is stored [in, out]: cps^T g, sum_rows g
        else:
            d_w2.copy_(cps.float().t() @ gb.float())
            d_barc.copy_(gb.float().sum(0))
        # ---- encoders ----
        d_pre = torch.empty((M, 3 * d), dtype=bf, device=dev)
        _C.check(lib.vlg_langfeat_split_backward(_C.ptr(d_txt), _C.BF16 if d_txt.dtype == bf else _C.F32, _C.ptr(d_child),
                                                 _C.ptr(d_parent), _C.ptr(d_sum), _C.BF16, _C.ptr(child), _C.ptr(parent), _C.ptr(heads),
                                                 B, N, d, slope, _C.ptr(d_pre), st), "langfeat_split_backward")
        if _wgrad_ok(M, 3 * d, h, bf):
            linear_wgrad(d_pre, x1, out=(d_wenc, d_benc))
        else:
            d_wenc.copy_(d_pre.float().t() @ x1.float())
            d_benc.copy_(d_pre.float().sum(0))
        d_x1 = d_pre @ w_enc_c                                                              # [M,h] bf16, library GEMM
        d_x = torch.empty((B, L, h), dtype=torch.float32, device=dev)
        _C.check(lib.vlg_langfeat_root_cat_backward(_C.ptr(d_x1), _C.ptr(lengths), B, L, h, _C.BF16, _C.ptr(d_x), st),
                 "langfeat_root_cat_backward")
        need = ctx.needs_input_grad
        pdt = (t_wenc, t_benc, t_w1, t_w2, t_barc)
        pg = [d_wenc, d_benc, d_w1, d_w2, d_barc]
        if all(t == pdt[0] for t in pdt) and pdt[0] != torch.float32:
            pg = outs.cast(5, pdt[0])
        else:
            pg = [g if g.dtype == t else g.to(t) for g, t in zip(pg, pdt)]
        return (d_x if d_x.dtype == t_x else d_x.to(t_x)) if need[0] else None, None, None, *(g if n else None for g, n in zip(pg, need[3:8])), None, None


def _wgrad2(x, dy):
    """(x^T dy [in,out], sum_rows dy [out]) for a weight stored [in, out] (`matmul(x, w2)`, joint.py:285)."""
    M = x.shape[0]
    if _wgrad_ok(M, dy.shape[1], x.shape[1], x.dtype) and dy.dtype == x.dtype:
        dwt, db = linear_wgrad(dy, x)               # [out, in]
        return dwt.t(), db
    return x.float().t() @ dy.float(), dy.float().sum(0)


def arc_word_features(x, lengths, heads, w_enc, b_enc, w1, w2, b_arc, slope=0.01, aux=None):
    """joint.py:262-288: txt = cat([word_encoder(x1), arc_repr]) with x1 = cat([masked mean, x]) -- [B,2N,d] bfloat16.

    x [B,L,h]; lengths [B] int64; heads [B,N] int64 (`predicted`, joint.py:256-258) or the handle of `start_structure` (joined
    inside, after the launches that do not need the heads);
    w_enc [3d,h] / b_enc [3d]: the word | child | parent encoders' Linear parameters concatenated along the output
    dimension (nn.Linear layout [out,in]; word: no activation, child / parent: LeakyReLU(slope) -- config/model/vlgae.yaml:69-73,
    joint.py:216-222); w1 [d,d,d], w2 [d,d], b_arc [d]: the arc encoder (joint.py:223-232).  Dropout is the identity (eval / p = 0).
    aux: optional dict that receives the child / parent activations [B,N,d] (inspection only)."""
    x = _plain(x)
    _C.require_gpu(x, "arc_word_features")
    B, L, h = x.shape
    d = w1.shape[0]
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
    return _LangFeat.apply(x, lengths.contiguous(), heads, w_enc, b_enc, w1, w2, b_arc, float(slope), aux)


def start_structure(merged_dec, merged_attach, lengths, keep_viterbi=False):
    """The two DPs of lang_feat_max_tree (joint.py:251-258) started on side streams as soon as the potentials exist -- before the
    attention-fuse that produces `x` -- so that they overlap it: pass the returned handle as `structure=` to lang_feat_max_tree."""
    import vlgae_amd.torch_struct as ts
    with torch.no_grad():
        return ts.DMV1o([merged_dec.detach(), merged_attach.detach()], lengths).marginals_and_heads_async(keep_viterbi)


def lang_feat_max_tree(x, lengths, merged_dec, merged_attach, w_enc, b_enc, w1, w2, b_arc, add_marginal=True, slope=0.01,
                       keep_viterbi=False, aux=None, structure=None):
    """`DependencyBoxRel.lang_feat_max_tree` (joint.py:235-292) -> (txt [B,2N,d] bf16, txt_mask [B,2N] bool, txt_marginal
    [B,2N] float32).  The potentials are constants of this stage (detached, joint.py:252-253).  `structure` = the handle of an
    earlier `start_structure(...)` (then merged_dec / merged_attach / keep_viterbi are not used here).
    Measured: the two DPs are joined BEFORE the root row and the projection GEMM.  Letting those launches run beside the DPs
    (they do not need the heads) made the training step 90 us SLOWER (1.39 -> 1.48 ms as one HIP graph, same box): each DP is one
    workgroup per CU on a 93 us critical path, and a workgroup that has to wait for a CU behind a GEMM tile lengthens that path."""
    import vlgae_amd.torch_struct as ts
    with torch.no_grad():
        if structure is not None:
            _, marg, heads = structure.wait()
        else:
            marg, heads = ts.DMV1o([merged_dec.detach(), merged_attach.detach()], lengths).marginals_and_heads(keep_viterbi)
        txt_marginal, txt_mask = txt_marginal_and_mask(marg, heads, lengths, add_marginal)
    if aux is not None:
        aux["heads"] = heads
    txt = arc_word_features(x, lengths, heads, w_enc, b_enc, w1, w2, b_arc, slope, aux)
    return txt, txt_mask, txt_marginal
