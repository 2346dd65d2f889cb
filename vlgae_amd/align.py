"""Region x word alignment -- host-side mirror of the reference's interface for this half of the path.

    gather_logit_simple(inputs, vis, txt, vp)   <->  DependencyBoxRel.gather_logit_simple (joint.py:406-419)
    attention_fuse(vis_feat, txt_feat, vis_mid, enc_x, layernorm)  <->  joint.py:670-674

`gather_logit_simple` keeps the reference's calling convention (tuples of (feat, mask, extra), named
tensors in, named ('B','A','Q','V') tensor out) so it can be registered under the reference's impl
registry unchanged:  JointModelBase.add_impl_to_group("gather_logit", "mi355x")(gather_logit_simple)
(base.py:118-142; selected by `gather_logit_mode` in config/model/vlgae.yaml:57).
"""
import functools
import math
import os

import torch
from torch.autograd.function import once_differentiable

from . import _C

INF = 1e20  # src/__init__.py:110 -- the pipeline's fill value for masked logits


def _plain(t):
    return t.rename(None) if t is not None and any(n is not None for n in t.names) else t


def bilinear_align(txt_feat, vis_feat, txt_mask=None, vis_mask=None, neg_inf=-INF, full=True, max_v=False,
                   max_q=False, diag=False):
    """attmap[b,a,q,v] = <txt[b,q], vis[a,v]>, masked to `neg_inf`; optionally the fused reductions the
    grounding loss / decoder consume (joint.py:473-483, :519-524) without materialising [B,A,Q,V].

    Returns a dict with the requested entries among full [B,A,Q,V], max_v [B,A,Q], max_q [B,A,V],
    diag [B,Q,V]; all float32."""
    txt_feat, vis_feat, txt_mask, vis_mask = _plain(txt_feat), _plain(vis_feat), _plain(txt_mask), _plain(vis_mask)
    _C.require_gpu(txt_feat, "bilinear_align")
    B, Q, d = txt_feat.shape
    A, V, d2 = vis_feat.shape
    if d != d2:
        raise ValueError(f"feature dims differ: txt {d} vs vis {d2}")
    if vis_feat.dtype != txt_feat.dtype:
        vis_feat = vis_feat.to(txt_feat.dtype)
    dt, txt_c = _C.in_dtype(txt_feat)
    _, vis_c = _C.in_dtype(vis_feat)
    dev = txt_feat.device
    tm = _C.mask_u8(txt_mask, dev)
    vm = _C.mask_u8(vis_mask, dev)
    if tm is not None and tuple(tm.shape) != (B, Q):
        raise ValueError(f"txt_mask must be [B,Q]={(B, Q)}, got {tuple(tm.shape)}")
    if vm is not None and tuple(vm.shape) != (A, V):
        raise ValueError(f"vis_mask must be [A,V]={(A, V)}, got {tuple(vm.shape)}")
    o_full = torch.empty((B, A, Q, V), dtype=torch.float32, device=dev) if full else None
    o_maxv = torch.empty((B, A, Q), dtype=torch.float32, device=dev) if max_v else None
    o_maxq = torch.empty((B, A, V), dtype=torch.float32, device=dev) if max_q else None
    o_diag = torch.empty((B, Q, V), dtype=torch.float32, device=dev) if diag else None
    _C.check(_C.lib().vlg_bilinear_align(_C.ptr(txt_c), _C.ptr(vis_c), _C.ptr(tm), _C.ptr(vm), B, A, Q, V, d, dt,
                                         float(neg_inf), _C.ptr(o_full), _C.ptr(o_maxv), _C.ptr(o_maxq),
                                         _C.ptr(o_diag), _C.stream_of(txt_feat)), "bilinear_align")
    return {k: v for k, v in (("full", o_full), ("max_v", o_maxv), ("max_q", o_maxq), ("diag", o_diag)) if v is not None}


class _GatherLogit(torch.autograd.Function):
    """attmap with gradients to both feature tensors (the grounding loss back-propagates through it).
    Backward of the contraction is two more contractions (vlg_bilinear_align_backward); masked entries carry no
    gradient (masked_fill_, joint.py:417-418)."""

    @staticmethod
    def forward(ctx, txt_feat, vis_feat, txt_mask, vis_mask, neg_inf):
        if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) and txt_feat.shape[-1] not in (32, 64, 128):
            # the backward kernels cover the matching widths 32 / 64 / 128: say so HERE, not inside loss.backward()
            raise ValueError(f"gather_logit: gradients need a matching width of 32, 64 or 128 (got d={txt_feat.shape[-1]}); "
                             "detach the features for inference-only use")
        ctx.save_for_backward(txt_feat, vis_feat, txt_mask, vis_mask)
        return bilinear_align(txt_feat, vis_feat, txt_mask, vis_mask, neg_inf)["full"]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        txt_feat, vis_feat, txt_mask, vis_mask = ctx.saved_tensors
        g_txt, g_vis = bilinear_align_backward(g, txt_feat, vis_feat, txt_mask, vis_mask, ctx.needs_input_grad[0],
                                               ctx.needs_input_grad[1])
        return (None if g_txt is None else g_txt.to(txt_feat.dtype), None if g_vis is None else g_vis.to(vis_feat.dtype),
                None, None, None)


def bilinear_align_backward(grad_out, txt_feat, vis_feat, txt_mask=None, vis_mask=None, want_txt=True, want_vis=True):
    """Gradients of `bilinear_align(...)["full"]` w.r.t. both feature tensors for the cotangent grad_out [B,A,Q,V] (what
    autograd derives for joint.py:413-418; masked positions pass no gradient).  Returns (g_txt [B,Q,d], g_vis [A,V,d]) in
    float32 (None where not wanted).  One HIP kernel per gradient reads the cotangent in place -- no permuted, masked or
    up-cast copies of it -- on the matrix cores (d = 128 at config-2 widths: bf16 MFMA with the cotangent, and fp32 features,
    split into bf16 terms -- relative error < 2^-16 per product; other shapes: exact fp32 MFMA); no library-GEMM / eager fallback."""
    txt_feat, vis_feat, txt_mask, vis_mask = _plain(txt_feat), _plain(vis_feat), _plain(txt_mask), _plain(vis_mask)
    _C.require_gpu(txt_feat, "bilinear_align_backward")
    B, Q, d = txt_feat.shape
    A, V, _ = vis_feat.shape
    if tuple(grad_out.shape) != (B, A, Q, V):
        raise ValueError(f"grad_out must be [B,A,Q,V]={(B, A, Q, V)}, got {tuple(grad_out.shape)}")
    dev = txt_feat.device
    g = _plain(grad_out)
    if g.dtype != torch.float32 or not g.is_contiguous():
        g = g.to(torch.float32).contiguous()
    dt, txt_c = _C.in_dtype(txt_feat)
    vis_c = vis_feat.detach().to(txt_c.dtype).contiguous()
    tm = _C.mask_u8(txt_mask, dev)
    vm = _C.mask_u8(vis_mask, dev)
    if want_txt or want_vis:
        nbytes = _C.lib().vlg_bilinear_align_backward_workspace(B, A, Q, V, d, dt)
        (g_txt, g_vis), ws = _C.alloc_f32(dev, ((B, Q, d) if want_txt else None, (A, V, d) if want_vis else None), nbytes)
        _C.check(_C.lib().vlg_bilinear_align_backward(_C.ptr(g), _C.ptr(txt_c), _C.ptr(vis_c), _C.ptr(tm), _C.ptr(vm), B, A, Q, V,
                                                      d, dt, _C.ptr(ws) if nbytes else None, nbytes, _C.ptr(g_txt), _C.ptr(g_vis),
                                                      _C.stream_of(txt_c)),
                 "bilinear_align_backward")
        return g_txt, g_vis
    return None, None


def gather_logit(inputs, vis, txt, vp=None):
    """attmap for every caption x image pair.  vis = (feat[A,V,d], mask[A,V], _), txt = (feat[B,Q,d],
    mask[B,Q], marginal); returns the named tensor ('B','A','Q','V') with -INF at masked positions."""
    vis_feat, vis_mask, _ = vis
    txt_feat, txt_mask, _ = txt
    out = _GatherLogit.apply(_plain(txt_feat), _plain(vis_feat), _plain(txt_mask), _plain(vis_mask), -INF)
    return out.refine_names("B", "A", "Q", "V")


def gather_logit_simple(self, inputs, vis, txt, vp):
    """Exactly the reference method's signature (joint.py:406-407), so it registers as an impl:
    `JointModelBase.add_impl_to_group("gather_logit", "mi355x")(gather_logit_simple)`."""
    return gather_logit(inputs, vis, txt, vp)


class _GatherLogitReduced(torch.autograd.Function):
    """gather_logit_reduced (joint.py:421-432) for B captions x B images: [B,B] marginal-weighted mean over the queries of
    the max over regions, with gradients to both feature tensors through the arg-max positions.  The marginal is a
    constant (joint.py:251-268 builds it from detached scores)."""

    @staticmethod
    def forward(ctx, txt_feat, vis_feat, txt_mask, vis_mask, marginal, neg_inf):
        dt, txt_c = _C.in_dtype(txt_feat)
        vis_c = vis_feat.detach().to(txt_c.dtype).contiguous()
        B, Q, d = txt_c.shape
        V = vis_c.shape[1]
        dev = txt_c.device
        tm = _C.mask_u8(txt_mask, dev)
        vm = _C.mask_u8(vis_mask, dev)
        marg = marginal.detach().to(device=dev, dtype=torch.float32).contiguous()
        nbytes = _C.lib().vlg_align_reduced_workspace(B, Q)
        (logit,), ws = _C.alloc_f32(dev, ((B, B),), nbytes)
        _C.check(_C.lib().vlg_align_reduced(_C.ptr(txt_c), _C.ptr(vis_c), _C.ptr(tm), _C.ptr(vm), _C.ptr(marg), B, Q, V, d, dt,
                                            float(neg_inf), _C.ptr(ws), nbytes, _C.ptr(logit), _C.stream_of(txt_c)),
                 "align_reduced")
        ctx.save_for_backward(txt_c, vis_c, tm, vm, marg, ws)
        ctx.meta = (dt, nbytes, txt_feat.dtype, vis_feat.dtype)
        return logit

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        txt_c, vis_c, tm, vm, marg, ws = ctx.saved_tensors
        dt, nbytes, t_dtype, v_dtype = ctx.meta
        B, Q, d = txt_c.shape
        V = vis_c.shape[1]
        g = g.to(torch.float32).contiguous()
        (g_txt, g_vis), _ = _C.alloc_f32(txt_c.device, ((B, Q, d) if ctx.needs_input_grad[0] else None,
                                                        (B, V, d) if ctx.needs_input_grad[1] else None))
        if g_txt is not None or g_vis is not None:
            _C.check(_C.lib().vlg_align_reduced_backward(_C.ptr(txt_c), _C.ptr(vis_c), _C.ptr(tm), _C.ptr(vm), _C.ptr(marg),
                                                         _C.ptr(g), B, Q, V, d, dt, _C.ptr(ws), nbytes, _C.ptr(g_txt),
                                                         _C.ptr(g_vis), _C.stream_of(txt_c)), "align_reduced_backward")
        return (None if g_txt is None else g_txt.to(t_dtype), None if g_vis is None else g_vis.to(v_dtype), None, None, None,
                None)


def gather_logit_reduced(self, inputs, vis, txt, vp):
    """The reference method's signature (joint.py:421-422), so it registers as an impl:
    `JointModelBase.add_impl_to_group("gather_logit", "reduced|mi355x")(gather_logit_reduced)`.
    vis = (feat[B,V,d], mask[B,V], _), txt = (feat[B,Q,d], mask[B,Q], marginal[B,Q]); returns logit [B(captions), B(images)],
    the input of loss_grounding_cap_img_ll (:493-499) and decode_grounding_on_image (:506-510), which stay as they are."""
    vis_feat, vis_mask, _ = vis
    txt_feat, txt_mask, txt_marginal = txt
    txt_feat, vis_feat, txt_mask, vis_mask, txt_marginal = (_plain(t) for t in (txt_feat, vis_feat, txt_mask, vis_mask,
                                                                                txt_marginal))
    _C.require_gpu(txt_feat, "gather_logit_reduced")
    B, Q, d = txt_feat.shape
    if vis_feat.shape[0] != B or vis_feat.shape[2] != d:
        raise ValueError(f"gather_logit_reduced pairs B captions with B images: txt {tuple(txt_feat.shape)} vis {tuple(vis_feat.shape)}")
    if tuple(txt_marginal.shape) != (B, Q):
        raise ValueError(f"txt_marginal must be [B,Q]={(B, Q)}, got {tuple(txt_marginal.shape)}")
    return _GatherLogitReduced.apply(txt_feat, vis_feat, txt_mask, vis_mask, txt_marginal, -INF)


@functools.lru_cache(maxsize=256)
def _attn_sizes(B, L, V, d, h, key_chunk):
    """(forward workspace, saved-record bytes, adjoint workspace fp32 grads, adjoint workspace bf16 grads): pure functions of the shape --
    asked once per shape (four ctypes calls per step are host time an eager step does not have)."""
    lib = _C.lib()
    return (lib.vlg_attn_fuse_workspace(B, L, V, h, key_chunk), lib.vlg_attn_fuse_saved_bytes(B, L, V, h, key_chunk),
            lib.vlg_attn_fuse_backward_workspace(B, L, V, d, h, _C.F32, key_chunk), lib.vlg_attn_fuse_backward_workspace(B, L, V, d, h, _C.BF16, key_chunk))


def _attn_fuse_launch(vis_c, txt_c, mid_c, enc_c, gamma, beta, eps, dt, want_att, key_chunk=0, save=False):
    B, V, d = vis_c.shape
    L, h = txt_c.shape[1] - 1, mid_c.shape[2]
    lib = _C.lib()
    sizes = _attn_sizes(B, L, V, d, h, key_chunk)
    nbytes = 0 if want_att else sizes[0]                     # chunk records of the key-split form (many keys)
    sbytes = sizes[1] if save and not want_att else 0
    (out, att), ws = _C.alloc_f32(vis_c.device, ((B, L, h), (B, L, V) if want_att else None), nbytes)
    # the merged records of the key-split form, kept for the adjoint (a tensor of its own: the chunk records above are ~8x its size)
    saved = torch.empty(sbytes // 4, dtype=torch.float32, device=vis_c.device) if sbytes else None
    _C.check(lib.vlg_attn_fuse(_C.ptr(vis_c), _C.ptr(txt_c), _C.ptr(mid_c), _C.ptr(enc_c), _C.ptr(gamma), _C.ptr(beta), B, L, V, d, h, dt,
                               float(eps), key_chunk, _C.ptr(ws) if nbytes else None, nbytes, _C.ptr(saved), _C.ptr(att), _C.ptr(out),
                               _C.stream_of(vis_c)), "attn_fuse")
    return (out, att, saved) if save else (out, att)


class _AttnFuse(torch.autograd.Function):
    """joint.py:670-674 with gradients to the four feature tensors and the LayerNorm parameters (the fuse sits in
    DependencyBoxRel._forward, so the parser's loss back-propagates through it).  The adjoint recomputes the forward
    per 16-word tile instead of saving the attention map."""

    @staticmethod
    def forward(ctx, vis_feat, txt_feat, vis_mid, enc_x, ln_weight, ln_bias, eps, key_chunk):
        dt, vis_c = _C.in_dtype(vis_feat)
        txt_c, mid_c, enc_c = (t.detach().to(vis_c.dtype).contiguous() for t in (txt_feat, vis_mid, enc_x))
        gamma = ln_weight.detach().to(torch.float32).contiguous()
        beta = ln_bias.detach().to(torch.float32).contiguous()
        out, _, saved = _attn_fuse_launch(vis_c, txt_c, mid_c, enc_c, gamma, beta, eps, dt, False, key_chunk, save=True)
        ctx.save_for_backward(vis_c, txt_c, mid_c, enc_c, gamma)
        ctx.fwd_records = saved   # (many keys) the forward's merged streaming-softmax records: the adjoint does not recompute them
        ctx.meta = (dt, float(eps), key_chunk, vis_feat.dtype, txt_feat.dtype, vis_mid.dtype, enc_x.dtype, ln_weight.dtype, ln_bias.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        vis_c, txt_c, mid_c, enc_c, gamma = ctx.saved_tensors
        dt, eps, key_chunk, *dtypes = ctx.meta
        B, V, d = vis_c.shape
        L, h = txt_c.shape[1] - 1, mid_c.shape[2]
        dev = vis_c.device
        sb, sl, sh = dout.stride()
        if dout.dtype != torch.float32 or sh != 1 or sb % 4 or sl % 4 or dout.data_ptr() % 16:
            dout = dout.to(torch.float32).contiguous()     # (a broadcast over the positions -- stride 0 -- is read in place)
            sb, sl = L * h, h
        lib = _C.lib()
        # bf16 features throughout: the four feature gradients leave the kernels as bf16 (rounded once from the fp32 accumulators -- no
        # fp32 round trip through HBM, no cast launch; at V = 1369, B = 64 the fp32 d_vis_mid alone is 90 MB)
        bf = dt == _C.BF16 and dtypes[0] == dtypes[1] == dtypes[2] == dtypes[3] == torch.bfloat16
        gdt = _C.BF16 if bf else _C.F32
        nbytes = _attn_sizes(B, L, V, d, h, key_chunk)[3 if bf else 2]
        shapes = ((B, V, d), (B, L + 1, d), (B, V, h), (B, L, h))
        if bf:   # one bf16 allocation for the four feature gradients, one fp32 one for the affine pair + the scratch
            numels = [math.prod(sh_) for sh_ in shapes]
            padded = [(n + 127) & ~127 for n in numels]
            flat = torch.empty(sum(padded), dtype=torch.bfloat16, device=dev)
            feats, o = [], 0
            for sh_, n, pn in zip(shapes, numels, padded):
                feats.append(flat[o:o + n].view(sh_))
                o += pn
            (dg, db), ws = _C.alloc_f32(dev, ((h,), (h,)), nbytes)
            outs = feats + [dg, db]
        else:   # one allocation for the six gradients and the scratch (host overhead matters at these sizes)
            outs, ws = _C.alloc_f32(dev, shapes + ((h,), (h,)), nbytes)
        _C.check(lib.vlg_attn_fuse_backward(_C.ptr(vis_c), _C.ptr(txt_c), _C.ptr(mid_c), _C.ptr(enc_c), _C.ptr(gamma), _C.ptr(dout), sb, sl,
                                            B, L, V, d, h, dt, eps, key_chunk, gdt, _C.ptr(ctx.fwd_records), _C.ptr(ws), nbytes, *(_C.ptr(o) for o in outs),
                                            _C.stream_of(vis_c)), "attn_fuse_backward")
        grads = [(o if o.dtype == t else o.to(t)) if ctx.needs_input_grad[i] else None
                 for i, (o, t) in enumerate(zip(outs, dtypes))]
        return (*grads, None, None)


def attention_fuse(vis_feat, txt_feat, vis_mid, enc_x, ln_weight, ln_bias, eps=1e-5, return_attmap=False, key_chunk=0):
    """joint.py:670-674:  LayerNorm(enc_x + softmax_v(<vis, txt[:,1:]>) @ vis_mid).

    vis_feat [B,V,d], txt_feat [B,L+1,d] (root slot first), vis_mid [B,V,h], enc_x [B,L,h]; LayerNorm
    parameters [h].  Returns float32 [B,L,h]; differentiable in all six tensors (the adjoint kernels need d and h
    to be multiples of 16 and <= 256).  `return_attmap=True` also returns attmap [B,L,V] (inspection; no autograd).
    key_chunk: 0 = automatic -- one pass over the keys for V <= 256, the key-split kernels above that (the shipped factor layout has
    36 + 36^2 + 36 + 1 = 1369 keys per image: chunks of keys per wavefront, streaming-softmax records merged in chunk order, forward
    and adjoint; csrc/vlg_attn.hip); > 0 = that many keys per chunk (tests)."""
    vis_feat, txt_feat, vis_mid, enc_x = (_plain(t) for t in (vis_feat, txt_feat, vis_mid, enc_x))
    _C.require_gpu(vis_feat, "attention_fuse")
    B, V, d = vis_feat.shape
    L = txt_feat.shape[1] - 1
    h = vis_mid.shape[2]
    if tuple(txt_feat.shape) != (B, L + 1, d) or tuple(vis_mid.shape) != (B, V, h) or tuple(enc_x.shape) != (B, L, h):
        raise ValueError(f"attention_fuse: vis {tuple(vis_feat.shape)} txt {tuple(txt_feat.shape)} "
                         f"vis_mid {tuple(vis_mid.shape)} enc_x {tuple(enc_x.shape)}")
    tensors = (vis_feat, txt_feat, vis_mid, enc_x, ln_weight, ln_bias)
    if not return_attmap and torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        return _AttnFuse.apply(*tensors, float(eps), int(key_chunk))
    dt, vis_c = _C.in_dtype(vis_feat)
    txt_c, mid_c, enc_c = (t.detach().to(vis_c.dtype).contiguous() for t in (txt_feat, vis_mid, enc_x))
    gamma = ln_weight.detach().to(torch.float32).contiguous()
    beta = ln_bias.detach().to(torch.float32).contiguous()
    out, att = _attn_fuse_launch(vis_c, txt_c, mid_c, enc_c, gamma, beta, eps, dt, return_attmap, int(key_chunk))
    return (out, att) if return_attmap else out


# ----------------------------------------------------------------------------------------------
# Grounding loss on the fused alignment maxima (joint.py:439-491) -- no [B,A,Q,V] tensor
# ----------------------------------------------------------------------------------------------
def grounding_prior(tag, factor_names, vis_split, pos_for, Q, scale=100.0):
    """The additive POS prior of joint.py:446-470 (scale 100) / :528-552 (the decoder's, scale 1e10) as a table.  tag [B,L] (vp.tag); factor_names / vis_split as in
    `self.vis_factor_names` / `vis_packed[2]`; pos_for = {"obj": tensor, "rel": tensor, "attr": tensor}
    (`self.pos_for_*`).  Returns (pen [B,Q,S] float32, seg_of_v [V] uint8): region v of segment s loses pen[b,q,s]
    on the pair (b, b); rows outside the word queries 1..L stay 0."""
    B, L = tag.shape
    S = len(vis_split)
    dev = tag.device
    seg_of_v = torch.repeat_interleave(torch.arange(S, dtype=torch.uint8, device=dev),
                                       torch.as_tensor(list(vis_split), device=dev))
    pen = torch.zeros((B, Q, S), dtype=torch.float32, device=dev)
    for f, name in enumerate(factor_names):
        if name not in ("obj", "rel", "attr"):
            continue
        hit = tag.unsqueeze(-1).eq(pos_for[name].to(dev)).any(-1).to(torch.float32) * scale    # joint.py:452-463
        others = [s for s in range(S) if s != f]
        pen[:, 1:L + 1, others] += hit.unsqueeze(-1)
    return pen, seg_of_v


class _GroundingLoss(torch.autograd.Function):
    """total of loss_grounding_factor_ce with gradients to both feature tensors, computed in one pass: alignment
    maxima + arg-max (matrix cores) -> the two cross-entropies -> sparse row updates through the arg-max positions."""

    @staticmethod
    def forward(ctx, txt_feat, vis_feat, txt_mask, vis_mask, marginal, pen, seg_of_v, num_token, w_vis2txt, neg_inf):
        B, Q, d = txt_feat.shape
        V = vis_feat.shape[1]
        dt, txt_c = _C.in_dtype(txt_feat)
        vis_c = vis_feat.detach().to(txt_c.dtype).contiguous()
        dev = txt_c.device
        tm = _C.mask_u8(txt_mask, dev)
        vm = _C.mask_u8(vis_mask, dev)
        marg = marginal.detach().to(device=dev, dtype=torch.float32).contiguous()
        n_seg = 0
        if pen is not None:
            pen = pen.to(device=dev, dtype=torch.float32).contiguous()
            seg_of_v = seg_of_v.to(device=dev, dtype=torch.uint8).contiguous()
            n_seg = pen.shape[2]
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        nbytes = _C.lib().vlg_grounding_loss_workspace(B, Q, V)
        (sums, g_txt, g_vis), ws = _C.alloc_f32(dev, ((3,), (B, Q, d) if need else None, (B, V, d) if need else None), nbytes)
        _C.check(_C.lib().vlg_grounding_loss(_C.ptr(txt_c), _C.ptr(vis_c), _C.ptr(tm), _C.ptr(vm), _C.ptr(marg), _C.ptr(pen),
                                             _C.ptr(seg_of_v), n_seg, B, Q, V, d, dt, float(neg_inf), float(num_token),
                                             float(w_vis2txt), _C.ptr(ws), nbytes, _C.ptr(sums), _C.ptr(g_txt), _C.ptr(g_vis),
                                             _C.stream_of(txt_c)), "grounding_loss")
        ctx.save_for_backward(g_txt, g_vis)
        ctx.dtypes = (txt_feat.dtype, vis_feat.dtype)
        ctx.mark_non_differentiable(sums)
        ctx.set_materialize_grads(False)   # (no zero-fill launch for the cotangent `sums` never gets)
        return sums[2], sums   # (a 0-d view of the non-differentiable sums: no clone launch; nothing writes either in place)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_total, _g_sums):
        g_txt, g_vis = ctx.saved_tensors
        out = [None] * 10
        if g_total is None:
            return tuple(out)
        want_t, want_v = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        dt = ctx.dtypes[0]
        if dt == ctx.dtypes[1] and dt in (torch.float32, torch.bfloat16) and g_total.dtype == torch.float32 and g_total.numel() == 1:
            # scale by the upstream scalar and cast, both tensors in one launch
            a, b = (g_txt if want_t else None), (g_vis if want_v else None)
            oa = None if a is None else torch.empty(a.shape, dtype=dt, device=a.device)
            ob = None if b is None else torch.empty(b.shape, dtype=dt, device=b.device)
            if a is not None or b is not None:
                _C.check(_C.lib().vlg_scale_counts(_C.ptr(a), _C.ptr(b), _C.ptr(g_total), 0, 1, 0 if a is None else a.numel(),
                                                   0 if b is None else b.numel(), _C.BF16 if dt == torch.bfloat16 else _C.F32,
                                                   _C.ptr(oa), _C.ptr(ob), _C.stream_of(g_txt)), "scale_counts")
            out[0], out[1] = oa, ob
            return tuple(out)
        if want_t:
            out[0] = (g_txt * g_total).to(ctx.dtypes[0])
        if want_v:
            out[1] = (g_vis * g_total).to(ctx.dtypes[1])
        return tuple(out)


def grounding_loss_factor_ce(txt_feat, vis_feat, txt_mask, vis_mask, txt_marginal, num_token, vis2txt=1.0, pen=None,
                             seg_of_v=None, neg_inf=-INF):
    """gather_logit_simple + loss_grounding_factor_ce (joint.py:406-419, 439-491) for B captions x B images.

    Returns (total, sums) with sums = [txt2vis, vis2txt, total] (raw sums; the reference reports
    s / (s.detach() + 1e-6) * num_token for each, which total already contains).  `total` back-propagates to txt_feat
    and vis_feat; txt_marginal is a constant, as in the reference (joint.py:251-268 builds it from detached scores)."""
    txt_feat, vis_feat, txt_mask, vis_mask, txt_marginal = (_plain(t) for t in (txt_feat, vis_feat, txt_mask, vis_mask,
                                                                                txt_marginal))
    _C.require_gpu(txt_feat, "grounding_loss_factor_ce")
    B, Q, d = txt_feat.shape
    if vis_feat.shape[0] != B or vis_feat.shape[2] != d:
        raise ValueError(f"grounding loss pairs caption b with image b: txt {tuple(txt_feat.shape)} vis {tuple(vis_feat.shape)}")
    if tuple(txt_marginal.shape) != (B, Q):
        raise ValueError(f"txt_marginal must be [B,Q]={(B, Q)}, got {tuple(txt_marginal.shape)}")
    return _GroundingLoss.apply(txt_feat, vis_feat, txt_mask, vis_mask, txt_marginal, pen, seg_of_v, num_token, vis2txt,
                                neg_inf)


def loss_grounding_factor_ce(self, inputs, vp):
    """The reference method's signature (joint.py:441-442), so it registers as an impl:
    `JointModelBase.add_impl_to_group("loss_grounding", "factor|ce|mi355x")(loss_grounding_factor_ce)`.
    Reads the packed features instead of inputs["match_logit"] (which then never needs to be built); the entries of the
    returned dict are the reference's per-term values, detached (they are only logged there)."""
    txt_feat, txt_mask, txt_marginal = inputs["txt_packed"]
    vis_feat, vis_mask, vis_split = inputs["vis_packed"]
    args = self.cfg.loss_grounding_args
    pen = seg = None
    if args.use_pos_prior:
        pos_for = {"obj": self.pos_for_obj, "rel": self.pos_for_rel, "attr": self.pos_for_attr}
        pen, seg = grounding_prior(vp.tag, self.vis_factor_names, vis_split, pos_for, txt_feat.shape[1])
    num = vp.num_token
    total, sums = grounding_loss_factor_ce(txt_feat, vis_feat, txt_mask, vis_mask, txt_marginal, num, float(args.vis2txt),
                                           pen, seg)
    loss = {"txt2vis": sums[0] / (sums[0] + 1e-6) * num}
    if args.vis2txt > 0:
        loss["mt_vis2txt"] = args.vis2txt * sums[1] / (sums[1] + 1e-6) * num
    return total, loss


# ----------------------------------------------------------------------------------------------
# Grounding decoder on the fused alignment outputs (joint.py:512-629) -- no [B,A,Q,V] tensor
# ----------------------------------------------------------------------------------------------
def grounding_decode(txt_feat, vis_feat, txt_mask, vis_mask, pen=None, seg_of_v=None, use_heuristic=False, n_box=0,
                     rel_offset=-1, attr_offset=-1, n_word_rows=0, neg_inf=-INF, split_rows=True):
    """gather_logit_simple + the tensor half of decode_grounding_on_factor (joint.py:406-419, 516-596) for B captions x B
    images: the diagonal block and max over V come straight from the alignment kernel, then one launch applies the POS
    prior (`pen`, `seg_of_v` from grounding_prior(..., scale=1e10)) and the box heuristics and extracts the five best
    columns of every query row.  Returns dict(logit [B,Q,V] float32 (edited block), top5 [B,Q,5] int32 (descending,
    equal values by ascending column, -1 past V), factor2img [B,Q] int32).  `split_rows=False` keeps one workgroup per
    sentence even for small batches (same results; the kernel's single-launch form)."""
    r = bilinear_align(txt_feat, vis_feat, txt_mask, vis_mask, neg_inf, full=False, max_v=True, diag=True)
    logit, max_v = r["diag"], r["max_v"]
    B, Q, V = logit.shape
    if max_v.shape[1] != B:
        raise ValueError(f"grounding decode pairs caption b with image b: {B} captions, {max_v.shape[1]} images")
    dev = logit.device
    if pen is not None:
        pen = pen.to(device=dev, dtype=torch.float32).contiguous()
        seg_of_v = seg_of_v.to(device=dev, dtype=torch.uint8).contiguous()
        if pen.shape[:2] != (B, Q) or seg_of_v.numel() != V:
            raise ValueError(f"pen {tuple(pen.shape)} / seg_of_v {tuple(seg_of_v.shape)} do not match B={B} Q={Q} V={V}")
    top5 = torch.empty((B, Q, 5), dtype=torch.int32, device=dev)
    f2i = torch.empty((B, Q), dtype=torch.int32, device=dev)
    nbytes = _C.lib().vlg_grounding_decode_workspace(B, int(n_box)) if split_rows else 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    _C.check(_C.lib().vlg_grounding_decode(_C.ptr(logit), _C.ptr(pen), _C.ptr(seg_of_v), 0 if pen is None else pen.shape[2],
                                           B, Q, V, int(bool(use_heuristic)), int(n_box), int(rel_offset), int(attr_offset),
                                           int(n_word_rows), _C.ptr(max_v), B, _C.ptr(f2i), _C.ptr(top5), _C.ptr(ws), nbytes,
                                           _C.stream_of(logit)), "grounding_decode")
    return {"logit": logit, "top5": top5, "factor2img": f2i}


def _filter_list(data, mask):
    """src/utility/fn.py:143-151: keep the entries whose mask entry is true, recursively over nested lists."""
    if isinstance(mask[0], list):
        return [_filter_list(d, m) for d, m in zip(data, mask)]
    if isinstance(mask[0], int):
        return [d for d, m in zip(data, mask) if m]
    raise ValueError(f"Bad mask value: {mask}")


def decode_grounding_on_factor(self, inputs, vp):
    """The reference method's signature (joint.py:512-513), so it registers as an impl:
    `JointModelBase.add_impl_to_group("decode_grounding", "on_factor|mi355x")(decode_grounding_on_factor)`.
    Reads the packed features instead of inputs["match_logit"]; returns the same dict: txt_to_factor[b][query][k] =
    (factor name, box id) or ("rel", (box id, box id)) for the five best factors of every unmasked query, and
    txt_to_img[b][query] = the image whose best region matches the query best."""
    from bisect import bisect_left
    from itertools import accumulate
    txt_feat, txt_mask, _ = inputs["txt_packed"]
    vis_feat, vis_mask, vis_split = inputs["vis_packed"]
    args = self.cfg.decode_grounding_args
    names = list(self.vis_factor_names)
    vis_split = [int(w) for w in vis_split]
    pen = seg = None
    if args.use_pos_prior:                                            # joint.py:528-552
        pos_for = {"obj": self.pos_for_obj, "rel": self.pos_for_rel, "attr": self.pos_for_attr}
        pen, seg = grounding_prior(vp.tag, names, vis_split, pos_for, txt_feat.shape[1], scale=1e10)
    start = [0] + list(accumulate(vis_split))
    out = grounding_decode(txt_feat, vis_feat, txt_mask, vis_mask, pen, seg, bool(args.use_heuristic), vis_split[0],
                           start[names.index("rel")] if "rel" in names else -1,
                           start[names.index("attr")] if "attr" in names else -1, vp.mask.shape[1] + 1)
    match = out["top5"][..., :min(5, sum(vis_split))].tolist()         # the one host sync of the decoder, as in :596
    box_ids = vp.vis_box_index.tolist() if "vis_box_index" in vp else [list(range(200)) for _ in range(len(match))]
    processed = []
    for inst_match, box_index in zip(match, box_ids):                  # joint.py:598-622
        inst = []
        for candidates in inst_match:
            row = []
            for idx in candidates:
                group = bisect_left(start, idx)
                if start[group] != idx:
                    group -= 1
                name = names[group]
                idx -= start[group]
                row.append((name, (box_index[idx // vis_split[0]], box_index[idx % vis_split[0]]) if name == "rel"
                            else box_index[idx]))
            inst.append(row)
        processed.append(inst)
    keep = _plain(txt_mask).tolist()
    return {"txt_to_factor": _filter_list(processed, keep), "txt_to_img": _filter_list(out["factor2img"], keep)}


# ----------------------------------------------------------------------------------------------
# Arc encoder (lang_feat word+maxdep, joint.py:281-287)
# ----------------------------------------------------------------------------------------------
def _trilinear_launch(child_c, w_c, parent_c, dt):
    M, X = child_c.shape
    H, Y = w_c.shape[1], w_c.shape[2]
    nbytes = _C.lib().vlg_trilinear_workspace(M, X, H, Y, dt)
    (out,), ws = _C.alloc_f32(child_c.device, ((M, H),), nbytes)
    _C.check(_C.lib().vlg_trilinear_ws(_C.ptr(child_c), _C.ptr(w_c), _C.ptr(parent_c), M, X, H, Y, dt, _C.ptr(ws) if nbytes else None,
                                       nbytes, _C.ptr(out), _C.stream_of(child_c)), "trilinear")
    return out


class _ArcTrilinear(torch.autograd.Function):
    """einsum('bcx,xhy,bcy->bch', child, w1, parent) without the [B,C,H,Y] intermediate, differentiable in all three."""

    @staticmethod
    def forward(ctx, child, w1, parent):
        lead = child.shape[:-1]
        X, H, Y = w1.shape
        dt, child_c = _C.in_dtype(child.detach().reshape(-1, X))
        w_c = w1.detach().to(child_c.dtype).contiguous()
        parent_c = parent.detach().reshape(-1, Y).to(child_c.dtype).contiguous()
        out = _trilinear_launch(child_c, w_c, parent_c, dt)
        ctx.save_for_backward(child_c, w_c, parent_c)
        ctx.meta = (dt, lead, child.dtype, w1.dtype, parent.dtype)
        return out.reshape(*lead, H)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        child_c, w_c, parent_c = ctx.saved_tensors
        dt, lead, t_child, t_w, t_parent = ctx.meta
        M, X = child_c.shape
        H, Y = w_c.shape[1], w_c.shape[2]
        dev = child_c.device
        g = g.reshape(M, H)
        if g.dtype != torch.float32 or not g.is_contiguous():
            g = g.to(torch.float32).contiguous()
        need = ctx.needs_input_grad
        nbytes = _C.lib().vlg_trilinear_backward_workspace(M, X, H, Y, dt)
        (d_child, d_w, d_parent), ws = _C.alloc_f32(dev, ((M, X) if need[0] else None, (X, H, Y) if need[1] else None,
                                                          (M, Y) if need[2] else None), nbytes)
        _C.check(_C.lib().vlg_trilinear_backward(_C.ptr(child_c), _C.ptr(w_c), _C.ptr(parent_c), _C.ptr(g), M, X, H, Y, dt,
                                                 _C.ptr(ws), nbytes, _C.ptr(d_child), _C.ptr(d_w), _C.ptr(d_parent),
                                                 _C.stream_of(child_c)), "trilinear_backward")
        return (d_child.reshape(*lead, X).to(t_child) if need[0] else None, d_w.to(t_w) if need[1] else None,
                d_parent.reshape(*lead, Y).to(t_parent) if need[2] else None)


def arc_trilinear(child, w1, parent):
    """torch.einsum('bcx,xhy,bcy->bch', child, w1, parent) (joint.py:282-284); float32 result [..., H]."""
    child, parent = _plain(child), _plain(parent)
    _C.require_gpu(child, "arc_trilinear")
    X, H, Y = w1.shape
    if child.shape[-1] != X or parent.shape[-1] != Y or child.shape[:-1] != parent.shape[:-1]:
        raise ValueError(f"arc_trilinear: child {tuple(child.shape)} w1 {tuple(w1.shape)} parent {tuple(parent.shape)}")
    return _ArcTrilinear.apply(child, w1, parent)


def arc_encoder(child_repr, parent_repr, arc_encoder_w1, arc_encoder_w2, arc_encoder_b):
    """joint.py:281-287: the trilinear term on the matrix cores, the affine term as a plain library GEMM."""
    tri = arc_trilinear(child_repr, arc_encoder_w1, parent_repr)
    return tri + torch.matmul(_plain(child_repr) + _plain(parent_repr), arc_encoder_w2).float() + arc_encoder_b.float()


# ----------------------------------------------------------------------------------------------
# Encoder projections around the contraction (MLP, src/model/nn/common.py:23-51; joint.py:136-138,175,270-277)
# ----------------------------------------------------------------------------------------------
class WgradGroup:
    """Deferred split-K reductions: `linear_wgrad(..., defer=group)` issues the split-K launch only; `group.flush()` adds the partial tiles of
    every deferred product in ONE launch (up to 12 per launch; vlg_linear_wgrad_reduce_group) -- same fixed-order sums, same bits.  The
    outputs are valid after flush()."""

    def __init__(self, lazy=False):
        # lazy: the split-K launches wait for flush() too and go out as ONE grid per kernel image (vlg_linear_wgrad_partial_group): the weight
        # gradients of a backward pass are leaves, and a product of 256 workgroups launched alone pays the chip's fill and drain.  The operands
        # are referenced until then and must not be overwritten in place after linear_wgrad(..., defer=group) took them.
        self.items, self.keep, self.first, self.lazy, self.partials = [], [], None, lazy, []

    def flush(self):
        n = len(self.items)
        if n == 0:
            return
        if self.partials:
            parr = (_C.WgradPartial * len(self.partials))()
            for rec, vals in zip(parr, self.partials):
                rec.dy, rec.x, rec.ws, rec.ws_bytes, rec.ld_dy, rec.ld_x, rec.K, rec.M, rec.N, rec.in_dtype, rec.want_bias, rec.want_x_colsum = vals
            _C.check(_C.lib().vlg_linear_wgrad_partial_group(parr, len(self.partials), _C.stream_of(self.first)), "linear_wgrad_partial_group")
            self.partials = []
        arr = (_C.WgradReduce * n)()
        for rec, vals in zip(arr, self.items):
            rec.ws, rec.d_weight, rec.d_bias, rec.x_colsum, rec.K, rec.M, rec.N, rec.ld_dw, rec.out_dtype, rec.in_dtype = vals
        _C.check(_C.lib().vlg_linear_wgrad_reduce_group(arr, n, _C.stream_of(self.first)), "linear_wgrad_reduce_group")
        self.items, self.keep, self.first = [], [], None


def linear_wgrad(dy, x, want_bias=True, want_x_colsum=False, out=None, out_dtype=torch.float32, defer=None):
    """Weight / bias gradient of `y = x @ weight.T + bias` over all token rows: (dy^T x [out, in], sum_rows dy [out]), float32.
    want_x_colsum: the second result is sum_rows x [in] instead (a weight stored [in, out]: pass the layer input as dy and the
    cotangent as x).  out = (d_weight, second) writes into caller-owned tensors (both of one type); out_dtype (float32 / bfloat16):
    the type of the results when `out` is not given -- the parameter's own, so that no cast launch follows the reduction.

    dy [K, out], x [K, in]: both bf16 or both float32, row-major (row strides that are multiples of 8 elements are taken in place -- column
    slices of wider buffers), out and in multiples of 8 (64 x 64 output tiles; the last tile of either side may be partial).  Split over the rows across the whole chip, fixed summation
    order (vlg_linear_wgrad); other shapes / dtypes raise -- callers decide (see `_Linear.backward`).  defer: a WgradGroup (see there).
    float32 operands (the reference's `precision: 32`) run on the bf16 matrix cores as three products per pair, a_hi b_hi + a_hi b_lo + a_lo b_hi
    (hi = bf16(v), lo = bf16(v - hi)): <= ~2^-16 relative per product before the fp32 accumulation."""
    _C.require_gpu(dy, "linear_wgrad")
    K, M = dy.shape
    N = x.shape[1]
    if dy.dtype != x.dtype or dy.dtype not in (torch.bfloat16, torch.float32) or x.shape[0] != K:
        raise ValueError(f"linear_wgrad: [K,out] / [K,in] of one type (bf16 or float32) expected, got {dy.dtype} {tuple(dy.shape)} / {x.dtype} {tuple(x.shape)}")
    idt = _C.BF16 if dy.dtype == torch.bfloat16 else _C.F32
    if dy.stride(1) != 1 or dy.stride(0) % 8 or dy.data_ptr() % 16:
        dy = dy.contiguous()
    if x.stride(1) != 1 or x.stride(0) % 8 or x.data_ptr() % 16:
        x = x.contiguous()
    nbytes = _C.lib().vlg_linear_wgrad_workspace(K, M, N)
    if nbytes == 0:
        raise ValueError(f"linear_wgrad: unsupported shape K={K} out={M} in={N} (out, in must be multiples of 8)")
    second = (N,) if want_x_colsum else ((M,) if want_bias else None)
    if out is None:
        if out_dtype == torch.float32:
            (dw, db), ws = _C.alloc_f32(dy.device, ((M, N), second), nbytes)
        else:
            n2 = 0 if second is None else second[0]
            buf = torch.empty(M * N + n2, dtype=out_dtype, device=dy.device)
            dw, db = buf[:M * N].view(M, N), (buf[M * N:] if n2 else None)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    else:
        dw, db = out
        out_dtype = dw.dtype
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    if (out_dtype not in (torch.float32, torch.bfloat16) or (db is not None and db.dtype != out_dtype) or tuple(dw.shape) != (M, N)
            or dw.stride(1) != 1 or dw.stride(0) < N):
        raise ValueError(f"linear_wgrad: outputs must be float32 or bfloat16 of one type, d_weight [out, in] with unit column stride (a column "
                         f"block of a wider gradient is fine), got {dw.dtype} {tuple(dw.shape)} {dw.stride()} / {None if db is None else db.dtype}")
    odt = _C.BF16 if out_dtype == torch.bfloat16 else _C.F32
    if defer is not None:   # the split-K launch alone (or, a lazy group, not even that yet); the reduction joins the group's single launch
        if defer.lazy:
            defer.partials.append((dy.data_ptr(), x.data_ptr(), ws.data_ptr(), nbytes, dy.stride(0), x.stride(0), K, M, N, idt,
                                   int(db is not None and not want_x_colsum), int(db is not None and want_x_colsum)))
            defer.keep.append((dy, x))
        else:
            _C.check(_C.lib().vlg_linear_wgrad_partial(_C.ptr(dy), dy.stride(0), _C.ptr(x), x.stride(0), K, M, N, idt, _C.ptr(ws), nbytes,
                                                       int(db is not None and not want_x_colsum), int(db is not None and want_x_colsum), _C.stream_of(dy)),
                     "linear_wgrad_partial")
        dpt = lambda t: None if t is None else t.data_ptr()
        defer.items.append((ws.data_ptr(), dw.data_ptr(), None if want_x_colsum else dpt(db), dpt(db) if want_x_colsum else None, K, M, N, dw.stride(0), odt, idt))
        defer.keep.append((ws, dw, db))
        if defer.first is None:
            defer.first = dy
        return dw, db
    _C.check(_C.lib().vlg_linear_wgrad(_C.ptr(dy), dy.stride(0), _C.ptr(x), x.stride(0), K, M, N, idt, _C.ptr(ws), nbytes, odt,
                                       _C.ptr(dw), dw.stride(0), None if want_x_colsum else _C.ptr(db), _C.ptr(db) if want_x_colsum else None,
                                       _C.stream_of(dy)), "linear_wgrad")
    return dw, db


def _small_problem(a, b, out=None, alpha=1.0, bias=None, rank1=None, accumulate=False, out_dtype=None):
    """Validates one small product and returns (argument tuple in vlg_small_gemm's order without the stream, out tensor)."""
    adt, ash, bsh, ast, bst = a.dtype, a.shape, b.shape, a.stride(), b.stride()     # (host time matters: ~28 calls per training step)
    if adt != b.dtype or (adt != torch.bfloat16 and adt != torch.float32):
        raise ValueError(f"small_matmul: both operands bfloat16 or both float32, got {adt} / {b.dtype}")
    if not a.is_cuda:
        _C.require_gpu(a, "small_matmul")
    na, nb_ = len(ash), len(bsh)
    if na not in (2, 3) or nb_ not in (2, 3):
        raise ValueError(f"small_matmul: 2-D or batched 3-D operands, got {tuple(ash)} @ {tuple(bsh)}")
    Z = ash[0] if na == 3 else (bsh[0] if nb_ == 3 else 0)
    M, K = ash[-2], ash[-1]
    N = bsh[-1]
    if K != bsh[-2] or (na == 3 and nb_ == 3 and ash[0] != bsh[0]):
        raise ValueError(f"small_matmul: {tuple(ash)} @ {tuple(bsh)}")
    shape = (Z, M, N) if Z else (M, N)
    if out is None:
        if accumulate:
            raise ValueError("small_matmul: accumulate needs out")
        out = torch.empty(shape, dtype=out_dtype or adt, device=a.device)
    elif tuple(out.shape) != shape or out.stride(-1) != 1 or (out.dtype != torch.float32 and out.dtype != torch.bfloat16):
        raise ValueError(f"small_matmul: out must be {shape} float32 / bfloat16 with unit last stride, got {tuple(out.shape)} {out.dtype} {out.stride()}")
    ost = out.stride()

    def vec(t, n, name):
        if t is None:
            return None, 0
        if t.dtype != adt or t.shape[-1] != n or t.stride(-1) != 1 or t.dim() not in (1, 2) or (t.dim() == 2 and t.shape[0] != max(Z, 1)):
            raise ValueError(f"small_matmul: {name} must be [{n}] or [{max(Z, 1)},{n}] {adt} with unit last stride, got {tuple(t.shape)} {t.dtype}")
        return t, (t.stride(0) if t.dim() == 2 else 0)

    bias, sbias = vec(bias, N, "bias") if bias is not None else (None, 0)
    (u, su), (v, sv) = (vec(rank1[0], M, "u"), vec(rank1[1], N, "v")) if rank1 is not None else ((None, 0), (None, 0))
    bf = torch.bfloat16
    dp = lambda t: None if t is None else t.data_ptr()
    return (a.data_ptr(), ast[0] if na == 3 else 0, ast[-2], ast[-1], b.data_ptr(), bst[0] if nb_ == 3 else 0, bst[-2], bst[-1],
            out.data_ptr(), ost[0] if Z else 0, ost[-2], dp(bias), sbias, dp(u), su, dp(v), sv,
            Z if Z else 1, M, N, K, float(alpha), 1 if accumulate else 0, _C.BF16 if adt == bf else _C.F32,
            _C.BF16 if out.dtype == bf else _C.F32), out


def small_matmul(a, b, out=None, alpha=1.0, bias=None, rank1=None, accumulate=False, out_dtype=None):
    """alpha * a @ b (+ bias over the rows, + outer(u, v), + out) for products whose OUTPUT is small -- a few hundred rows and columns,
    optionally batched: weight-space products such as the folded bottleneck W1 W0 of `DMVSkipConnectEncoder` (nn/dmv_spec.py:52-54), which
    a library GEMM runs on one workgroup (15-28 us); here one wavefront per 32 x 32 tile (vlg_small_gemm).
    a [M,K] or [Z,M,K], b [K,N] or [Z,K,N] with ANY strides (transposes / slices are read in place), both bfloat16 (bf16 products) or both
    float32 (exact fp32 products), fp32 accumulation; bias [N] or [Z,N]; rank1 = (u [M] or [Z,M], v [N] or [Z,N]); out: a [.., M, N] tensor
    with unit last stride to write (or, accumulate=True, add) into; out_dtype defaults to the operands' dtype."""
    args, out = _small_problem(a, b, out, alpha, bias, rank1, accumulate, out_dtype)
    rc = _C.lib().vlg_small_gemm(*args, _C.stream_of(a))
    if rc:
        _C.check(rc, "small_gemm")
    return out


class SmallMatmulGroup:
    """Independent small products collected into ONE launch (vlg_small_gemm_group): `add(...)` takes small_matmul's arguments and returns
    the output tensor (allocated now, written by `launch()`); no problem of a group may read what another one writes.  One dependency
    level of the weight-space products of the parser's feed-forwards is one group (vlgae_amd/parser_ff.py)."""
    _ORDER = ("a", "sab", "sam", "sak", "b", "sbb", "sbk", "sbn", "c", "scb", "ldc", "bias", "sbias", "u", "su", "v", "sv", "batch", "M", "N", "K",
              "alpha", "accumulate", "in_dtype", "out_dtype")

    def __init__(self):
        self.problems, self.keep, self.first = [], [], None

    def add(self, a, b, out=None, alpha=1.0, bias=None, rank1=None, accumulate=False, out_dtype=None):
        args, out = _small_problem(a, b, out, alpha, bias, rank1, accumulate, out_dtype)
        self.problems.append(args)
        # the records hold raw addresses and the kernel runs at launch(): every operand stays referenced until then (a caller's
        # temporary -- `w.to(bf16)`, `.t()` of one -- would otherwise go back to the caching allocator and be handed to the next cast)
        self.keep.append((a, b, out, bias, rank1))
        if self.first is None:
            self.first = a
        return out

    def launch(self):
        n = len(self.problems)
        if n == 0:
            return
        arr = (_C.SmallGemm * n)()
        for rec, args in zip(arr, self.problems):
            for name, val in zip(self._ORDER, args):
                setattr(rec, name, val)
        _C.check(_C.lib().vlg_small_gemm_group(arr, n, _C.stream_of(self.first)), "small_gemm_group")
        self.problems, self.keep, self.first = [], [], None     # stream order protects the operands from here on (same stream)


_FP32_WGRAD_LIBRARY = bool(os.environ.get("VLGAE_WGRAD_FP32_LIBRARY"))   # opt-out: float32 weight gradients through the library's fp32 GEMM


def _wgrad_ok(K, M, N, dtype):
    """Whether the split-K kernel serves this weight gradient.  Accuracy contract for float32 operands (`precision: 32`): every product is
    three bf16 matrix-core products of hi / lo split operands (a_hi b_hi + a_hi b_lo + a_lo b_hi), ~2^-16 relative per product with fp32
    accumulation -- float32-level results (checked against float64 in tests), not bit-equal to an fp32-FMA GEMM.  A caller that wants the
    library's true fp32 GEMM sets VLGAE_WGRAD_FP32_LIBRARY=1 before importing this package (2-9x slower at these shapes)."""
    if dtype == torch.float32 and _FP32_WGRAD_LIBRARY:
        return False
    return dtype in (torch.bfloat16, torch.float32) and M % 8 == 0 and N % 8 == 0 and M >= 8 and N >= 8 and K >= 2048


_KN_LIBRARY = bool(os.environ.get("VLGAE_FF_LIBRARY"))   # opt-out, as parser_ff's


def linear_kn(x, w, out=None, rng=None, site=0, p=0.0):
    """x [rows, 256] @ w [256, n] -> [rows, n] (bf16): the input gradient of a Linear with 256 outputs (g @ weight) as ONE row-streaming launch
    with the weight block in registers (vlg_ff_linear_kn); w may be a column slice of a wider matrix.  The library ran these
    [10^4, 256] x [256, 800] products on 200 workgroups at ~0.8 TB/s of their own bytes (27 us where the bytes take ~8).
    rng (an encoders.DeviceRng) with (site, p): the result times the keep-mask `encoders.dropout(..., rng=rng, site=site)` draws over [rows, n] --
    the adjoint of Linear(Dropout(x)) in one launch."""
    rows, n = x.shape[0], w.shape[1]
    if out is None:
        out = torch.empty((rows, n), dtype=x.dtype, device=x.device)
    _C.check(_C.lib().vlg_ff_linear_kn(_C.ptr(x), x.stride(0), _C.ptr(w), w.stride(0), rows, n, None if rng is None else _C.ptr(rng.state), int(site), float(p),
                                       _C.ptr(out), out.stride(0), _C.stream_of(x)), "ff_linear_kn")
    return out


def linear_kn_ok(x, w):
    """Whether `x @ w` can take vlg_ff_linear_kn: bf16, 256 contraction channels, enough rows to fill the chip, strides it reads in place."""
    return (not _KN_LIBRARY and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and x.dim() == 2 and w.dim() == 2 and x.shape[1] == 256
            and w.shape[0] == 256 and x.shape[0] >= 2048 and w.shape[1] % 8 == 0 and w.shape[1] >= 8 and x.stride(1) == 1 and w.stride(1) == 1
            and x.stride(0) % 8 == 0 and x.stride(0) >= 256 and x.data_ptr() % 16 == 0)


class _Linear(torch.autograd.Function):
    """x @ weight.T + bias with the tall-skinny weight gradient on the split-K kernel; forward and the input gradient are
    library GEMMs, except the input gradient of a 256-output layer in bf16 (linear_kn)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        g2, x2 = g.reshape(-1, g.shape[-1]), x.reshape(-1, x.shape[-1])
        if need_x and linear_kn_ok(g2, weight):
            dx = linear_kn(g2, weight).reshape(x.shape)
        else:
            dx = (g2 @ weight).reshape(x.shape) if need_x else None
        dw = db = None
        if need_w or need_b:
            if g2.dtype == x2.dtype and _wgrad_ok(x2.shape[0], g2.shape[1], x2.shape[1], g2.dtype):
                dw, db = linear_wgrad(g2, x2, want_bias=need_b, out_dtype=weight.dtype)
                dw = dw if need_w else None
            else:   # small or oddly shaped: the library's GEMM is the right tool
                dw = (g2.t() @ x2) if need_w else None
                db = g2.sum(0) if need_b else None
        return dx, dw, db


def linear(x, weight, bias=None):
    """torch.nn.functional.linear for the encoder projections over all B*N token rows (weight [out, in] as nn.Linear
    stores it); differs from the stock op only in how the weight / bias gradients are computed (see linear_wgrad)."""
    return _Linear.apply(_plain(x), weight, bias)
