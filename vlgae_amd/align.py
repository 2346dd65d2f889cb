"""Region x word alignment -- host-side mirror of the reference's interface for this half of the path.

    gather_logit_simple(inputs, vis, txt, vp)   <->  DependencyBoxRel.gather_logit_simple (joint.py:406-419)
    attention_fuse(vis_feat, txt_feat, vis_mid, enc_x, layernorm)  <->  joint.py:670-674

`gather_logit_simple` keeps the reference's calling convention (tuples of (feat, mask, extra), named
tensors in, named ('B','A','Q','V') tensor out) so it can be registered under the reference's impl
registry unchanged:  JointModelBase.add_impl_to_group("gather_logit", "mi355x")(gather_logit_simple)
(base.py:118-142; selected by `gather_logit_mode` in config/model/vlgae.yaml:57).
"""
import torch

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
    dt, txt_c = _C.in_dtype(txt_feat.detach())
    _, vis_c = _C.in_dtype(vis_feat.detach())
    dev = txt_feat.device
    tm = None if txt_mask is None else txt_mask.to(device=dev, dtype=torch.uint8).contiguous()
    vm = None if vis_mask is None else vis_mask.to(device=dev, dtype=torch.uint8).contiguous()
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
    Backward of the contraction is two more contractions; masked entries carry no gradient
    (masked_fill_, joint.py:417-418)."""

    @staticmethod
    def forward(ctx, txt_feat, vis_feat, txt_mask, vis_mask, neg_inf):
        ctx.save_for_backward(txt_feat, vis_feat, txt_mask, vis_mask)
        return bilinear_align(txt_feat, vis_feat, txt_mask, vis_mask, neg_inf)["full"]

    @staticmethod
    def backward(ctx, g):
        txt_feat, vis_feat, txt_mask, vis_mask = ctx.saved_tensors
        g = g.to(torch.float32)
        if txt_mask is not None:
            g = g * txt_mask[:, None, :, None].to(g.dtype)
        if vis_mask is not None:
            g = g * vis_mask[None, :, None, :].to(g.dtype)
        B, A, Q, V = g.shape
        # plain library GEMMs (rocBLAS via torch): d(txt)[b,q,:] = sum_{a,v} g * vis ; d(vis)[a,v,:] = sum_{b,q} g * txt
        g_txt = g_vis = None
        if ctx.needs_input_grad[0]:
            g_txt = torch.matmul(g.permute(0, 2, 1, 3).reshape(B * Q, A * V), vis_feat.reshape(A * V, -1).float())
            g_txt = g_txt.reshape(B, Q, -1).to(txt_feat.dtype)
        if ctx.needs_input_grad[1]:
            g_vis = torch.matmul(g.permute(1, 3, 0, 2).reshape(A * V, B * Q), txt_feat.reshape(B * Q, -1).float())
            g_vis = g_vis.reshape(A, V, -1).to(vis_feat.dtype)
        return g_txt, g_vis, None, None, None


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


def _attn_fuse_launch(vis_c, txt_c, mid_c, enc_c, gamma, beta, eps, dt, want_att):
    B, V, d = vis_c.shape
    L, h = txt_c.shape[1] - 1, mid_c.shape[2]
    out = torch.empty((B, L, h), dtype=torch.float32, device=vis_c.device)
    att = torch.empty((B, L, V), dtype=torch.float32, device=vis_c.device) if want_att else None
    _C.check(_C.lib().vlg_attn_fuse(_C.ptr(vis_c), _C.ptr(txt_c), _C.ptr(mid_c), _C.ptr(enc_c), _C.ptr(gamma),
                                    _C.ptr(beta), B, L, V, d, h, dt, float(eps), _C.ptr(att), _C.ptr(out),
                                    _C.stream_of(vis_c)), "attn_fuse")
    return out, att


class _AttnFuse(torch.autograd.Function):
    """joint.py:670-674 with gradients to the four feature tensors and the LayerNorm parameters (the fuse sits in
    DependencyBoxRel._forward, so the parser's loss back-propagates through it).  The adjoint recomputes the forward
    per 16-word tile instead of saving the attention map."""

    @staticmethod
    def forward(ctx, vis_feat, txt_feat, vis_mid, enc_x, ln_weight, ln_bias, eps):
        dt, vis_c = _C.in_dtype(vis_feat.detach())
        txt_c, mid_c, enc_c = (t.detach().to(vis_c.dtype).contiguous() for t in (txt_feat, vis_mid, enc_x))
        gamma = ln_weight.detach().to(torch.float32).contiguous()
        beta = ln_bias.detach().to(torch.float32).contiguous()
        out, _ = _attn_fuse_launch(vis_c, txt_c, mid_c, enc_c, gamma, beta, eps, dt, False)
        ctx.save_for_backward(vis_c, txt_c, mid_c, enc_c, gamma)
        ctx.meta = (dt, float(eps), vis_feat.dtype, txt_feat.dtype, vis_mid.dtype, enc_x.dtype, ln_weight.dtype, ln_bias.dtype)
        return out

    @staticmethod
    def backward(ctx, dout):
        vis_c, txt_c, mid_c, enc_c, gamma = ctx.saved_tensors
        dt, eps, *dtypes = ctx.meta
        B, V, d = vis_c.shape
        L, h = txt_c.shape[1] - 1, mid_c.shape[2]
        dev = vis_c.device
        dout = dout.to(torch.float32).contiguous()
        nbytes = _C.lib().vlg_attn_fuse_backward_workspace(B, L, V, h)
        ws = torch.empty(max(nbytes, 4) // 4, dtype=torch.float32, device=dev)
        outs = [torch.empty(shape, dtype=torch.float32, device=dev)
                for shape in ((B, V, d), (B, L + 1, d), (B, V, h), (B, L, h), (h,), (h,))]
        _C.check(_C.lib().vlg_attn_fuse_backward(_C.ptr(vis_c), _C.ptr(txt_c), _C.ptr(mid_c), _C.ptr(enc_c), _C.ptr(gamma),
                                                 _C.ptr(dout), B, L, V, d, h, dt, eps, _C.ptr(ws), nbytes,
                                                 *(_C.ptr(o) for o in outs), _C.stream_of(vis_c)), "attn_fuse_backward")
        grads = [o.to(t) if ctx.needs_input_grad[i] else None for i, (o, t) in enumerate(zip(outs, dtypes))]
        return (*grads, None)


def attention_fuse(vis_feat, txt_feat, vis_mid, enc_x, ln_weight, ln_bias, eps=1e-5, return_attmap=False):
    """joint.py:670-674:  LayerNorm(enc_x + softmax_v(<vis, txt[:,1:]>) @ vis_mid).

    vis_feat [B,V,d], txt_feat [B,L+1,d] (root slot first), vis_mid [B,V,h], enc_x [B,L,h]; LayerNorm
    parameters [h].  Returns float32 [B,L,h]; differentiable in all six tensors (the adjoint kernels need d and h
    to be multiples of 16 and <= 256).  `return_attmap=True` also returns attmap [B,L,V] (inspection; no autograd)."""
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
        return _AttnFuse.apply(*tensors, float(eps))
    dt, vis_c = _C.in_dtype(vis_feat.detach())
    txt_c, mid_c, enc_c = (t.detach().to(vis_c.dtype).contiguous() for t in (txt_feat, vis_mid, enc_x))
    gamma = ln_weight.detach().to(torch.float32).contiguous()
    beta = ln_bias.detach().to(torch.float32).contiguous()
    out, att = _attn_fuse_launch(vis_c, txt_c, mid_c, enc_c, gamma, beta, eps, dt, return_attmap)
    return (out, att) if return_attmap else out
