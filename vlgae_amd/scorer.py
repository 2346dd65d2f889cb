"""Score construction feeding the DP -- host-side mirror of the tensor half of `DiscriminativeNDMV._forward`
(src/model/ldndmv.py:179-209) with the factorised-bilinear scorers of src/model/nn/dmv_spec.py:57-76.

    ndmv_potentials(x1, x2, y1, y2, root_rule, token, head_mask)  ->  (merged_dec [B,N,2,2,2], merged_attach [B,N,N,2])

where x1 = attach_scorer.project1(h_parent), x2 = attach_scorer.project2(h_child), y1 = dec_scorer.project1(h_parent),
y2 = dec_scorer.project2(h_dec) are the scorers' projected inputs (plain nn.Linear outputs: library GEMMs) and root_rule is
the root scorer's [T] log-softmax.  The reference builds attach_rule [B,L,T,2,2] = einsum + log_softmax over tokens, gathers
it by the sentence's tokens, selects directions with tril / triu masks, masks function-word heads, gathers the root scores
and merges; here that is one launch (and one adjoint launch + a fixed-order reduction) and the rule table never exists.
`discriminative_ndmv_potentials` is the same with the three scorer modules applied first (drop-in for ldndmv.py:179-209).
"""
import torch
from torch.autograd.function import once_differentiable

from . import _C

INF = 1e20   # src/__init__.py:110


def _rows(t, r, dtype):
    """(tensor, row stride in elements) for a [..., r] operand: rows of r contiguous values a constant stride apart are taken in
    place (a column slice of a wider row-major buffer); anything else is made contiguous."""
    t = t.detach()
    if t.dtype != dtype:
        t = t.to(dtype)
    n = t.numel() // r
    if t.stride(-1) == 1 and n > 0:
        ld = t.stride(-2) if t.dim() > 1 else r
        flat = t.as_strided((n, r), (ld, 1), t.storage_offset()) if ld >= r else None
        # the view [n, r] with stride ld addresses the same elements iff the leading dimensions collapse onto one row index
        want, acc = [], ld
        for size in reversed(t.shape[:-1]):
            want.append(acc)
            acc *= size
        if flat is not None and list(reversed(want)) == list(t.stride()[:-1]):
            return t, ld
    return t.contiguous(), r


def _side_by_side(a, lda, b, ldb, r):
    """b's rows start r elements behind a's in one [rows, 2r] buffer (two projections out of one GEMM)?"""
    return lda == ldb == 2 * r and a.dtype == b.dtype and b.data_ptr() == a.data_ptr() + r * a.element_size()


class _NdmvPotentials(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2, y1, y2, root_rule, token, head_mask, mask_fill, out_dtype):
        B, L, _, _, r = x1.shape
        T = x2.shape[0]
        N = L + 1
        st = x1.dtype if x1.dtype in (torch.float32, torch.bfloat16) else torch.float32
        dt = _C.BF16 if st == torch.bfloat16 else _C.F32
        (x1_c, l1), (x2_c, l2), (y1_c, l3), (y2_c, l4) = (_rows(t, r, st) for t in (x1, x2, y1, y2))
        root_c = root_rule.detach().to(torch.float32).contiguous()
        hm = _C.mask_u8(head_mask, x1_c.device)
        dev = x1.device
        md = torch.empty((B, N, 2, 2, 2), dtype=out_dtype, device=dev)
        ma = torch.empty((B, N, N, 2), dtype=out_dtype, device=dev)
        _C.check(_C.lib().vlg_ndmv_potentials(_C.ptr(x1_c), l1, _C.ptr(x2_c), l2, _C.ptr(y1_c), l3, _C.ptr(y2_c), l4, _C.ptr(root_c),
                                              _C.ptr(token), _C.ptr(hm), B, L, T, r, dt, float(mask_fill),
                                              _C.BF16 if out_dtype == torch.bfloat16 else _C.F32, _C.ptr(md), _C.ptr(ma),
                                              _C.stream_of(x1)), "ndmv_potentials")
        ctx.save_for_backward(x1_c, x2_c, y1_c, y2_c, token, hm)
        ctx.meta = (B, L, T, r, dt, (l1, l2, l3, l4), x1.dtype, x2.dtype, y1.dtype, y2.dtype, root_rule.dtype)
        return md, ma

    @staticmethod
    @once_differentiable
    def backward(ctx, g_md, g_ma):
        x1_c, x2_c, y1_c, y2_c, token, hm = ctx.saved_tensors
        B, L, T, r, dt, (l1, l2, l3, l4), *dtypes = ctx.meta
        dev = x1_c.device
        g_md, g_ma = (g if g.dtype == torch.float32 and g.is_contiguous() else g.to(torch.float32).contiguous() for g in (g_md, g_ma))
        nbytes = _C.lib().vlg_ndmv_potentials_backward_workspace(B, L, T, r)
        # gradients leave in the storage type of the layers that receive them (one type for the four: else fp32 + a cast each)
        gdt = dtypes[0] if dtypes[0] in (torch.float32, torch.bfloat16) and all(t == dtypes[0] for t in dtypes[:4]) else torch.float32
        (d_root,), ws = _C.alloc_f32(dev, ((T,),), nbytes)
        if _side_by_side(x1_c, l1, y1_c, l3, r):   # the cotangents of two slices of one buffer, side by side again: no cat downstream
            both = torch.empty((B * L * 4, 2 * r), dtype=gdt, device=dev)
            d_x1, d_y1, ldd = both[:, :r].view(B, L, 2, 2, r), both[:, r:].view(B, L, 2, 2, r), 2 * r
        else:
            d_x1, d_y1, ldd = torch.empty((B, L, 2, 2, r), dtype=gdt, device=dev), torch.empty((B, L, 2, 2, r), dtype=gdt, device=dev), r
        small = torch.empty((T * 4 + 8, r), dtype=gdt, device=dev)
        d_x2, d_y2 = small[:T * 4].view(T, 2, 2, r), small[T * 4:].view(2, 2, 2, r)
        _C.check(_C.lib().vlg_ndmv_potentials_backward(_C.ptr(x1_c), l1, _C.ptr(x2_c), l2, _C.ptr(y1_c), l3, _C.ptr(y2_c), l4, _C.ptr(token),
                                                       _C.ptr(hm), _C.ptr(g_md), _C.ptr(g_ma), B, L, T, r, dt, _C.ptr(ws), nbytes,
                                                       _C.BF16 if gdt == torch.bfloat16 else _C.F32, _C.ptr(d_x1), ldd, _C.ptr(d_x2),
                                                       _C.ptr(d_y1), ldd, _C.ptr(d_y2), _C.ptr(d_root), _C.stream_of(x1_c)),
                 "ndmv_potentials_backward")
        grads = [g if g.dtype == t else g.to(t) for g, t in zip((d_x1, d_x2, d_y1, d_y2, d_root), dtypes)]
        return (*(g if n else None for g, n in zip(grads, ctx.needs_input_grad[:5])), None, None, None, None)


def ndmv_potentials(x1, x2, y1, y2, root_rule, token, head_mask=None, mask_fill=-INF, out_dtype=torch.float32):
    """ldndmv.py:185-209 from the scorers' projected inputs; see the module docstring.  x1, y1 [B,L,2,2,r]; x2 [T,2,2,r];
    y2 [2,2,2,r]; root_rule [T]; token [B,L] int64 in [0,T); head_mask [B,L] bool or None.  Returns the root-merged
    (dec [B,L+1,2,2,2], attach [B,L+1,L+1,2]) in `out_dtype` (float32 like `DMV1o.merge`, or bfloat16 storage for the DP)."""
    _C.require_gpu(x1, "ndmv_potentials")
    if x1.dim() != 5 or tuple(x1.shape[2:4]) != (2, 2):
        raise ValueError(f"ndmv_potentials: x1 must be [B,L,2,2,r], got {tuple(x1.shape)}")
    B, L, _, _, r = x1.shape
    T = x2.shape[0]
    if tuple(x2.shape) != (T, 2, 2, r) or tuple(y1.shape[:4]) != (B, L, 2, 2) or tuple(y2.shape) != (2, 2, 2, y1.shape[4]) or y1.shape[4] != r:
        raise ValueError(f"ndmv_potentials: x2 {tuple(x2.shape)} y1 {tuple(y1.shape)} y2 {tuple(y2.shape)} for x1 {tuple(x1.shape)} "
                         "(attach and dec scorers must share the rank r)")
    if tuple(root_rule.shape) != (T,) or tuple(token.shape) != (B, L) or token.dtype != torch.int64:
        raise ValueError(f"ndmv_potentials: root_rule {tuple(root_rule.shape)} (need [{T}]), token {tuple(token.shape)} {token.dtype}")
    if out_dtype not in (torch.float32, torch.bfloat16):
        raise ValueError("ndmv_potentials: out_dtype must be float32 or bfloat16")
    return _NdmvPotentials.apply(x1, x2, y1, y2, root_rule, token.contiguous(), head_mask, float(mask_fill), out_dtype)


def discriminative_ndmv_potentials(attach_scorer, dec_scorer, root_scorer, h_parent, h_child, h_root, h_dec, token, head_mask=None,
                                   mask_fill=-INF, out_dtype=torch.float32):
    """ldndmv.py:184-209 given the reference's three `DMVFactorizedBilinear` modules and the mid_ff outputs
    h_parent [B,L,2,2,H], h_child [1,T,2,2,H], h_root [1,1,2,2,H], h_dec [1,2,2,2,H].
    Returns dict(merged_dec, merged_attach, root_rule [B,T]) -- the entries of `out` the trained model reads after the
    rule-supervised initialisation epochs (during those, `attach_rule` itself is a loss input: use the reference's lines)."""
    if attach_scorer.r != dec_scorer.r:
        raise ValueError("the fused path needs attach_rank == dec_rank (config/model/vlgae.yaml:115-116 sets both to _rank)")
    x1, x2 = attach_scorer.project1(h_parent), attach_scorer.project2(h_child)[0]          # nn/dmv_spec.py:67-68
    y1, y2 = dec_scorer.project1(h_parent), dec_scorer.project2(h_dec)[0]
    root_rule = root_scorer(h_root, h_child).sum([-1, -2]).log_softmax(-1).squeeze(1)      # [1,T], ldndmv.py:205
    md, ma = ndmv_potentials(x1, x2, y1, y2, root_rule[0], token, head_mask, mask_fill, out_dtype)
    return {"merged_dec": md, "merged_attach": ma, "root_rule": root_rule.expand(h_parent.shape[0], -1)}
