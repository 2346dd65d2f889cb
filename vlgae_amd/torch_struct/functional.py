"""Autograd-aware entry points over the C ABI (include/vlgae_amd.h).

The reference differentiates its inside loop with autograd (hundreds of AsStridedBackward nodes per
call).  Here the outside pass is part of the same kernel launch as the inside
pass: when any potential requires grad, forward() runs the fused inside+outside kernel and keeps the
unit-upstream expected counts; backward() only scales them by the incoming gradient of logZ.
"""
import threading
import warnings

import torch
from torch.autograd.function import once_differentiable

from .. import _C
from .semirings import NEGINF


_WS_BYTES = {}


def _workspace(op, B, N, semiring, device):
    key = (op, B, N, semiring)
    nbytes = _WS_BYTES.get(key)
    if nbytes is None:
        nbytes = _WS_BYTES[key] = _C.lib().vlg_workspace_bytes(op, B, N, semiring)
    if nbytes == 0:
        return None, 0
    return torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def _lengths(lengths, B, device, allow_none=False):
    if lengths is None:
        if allow_none:
            return None
        raise ValueError("lengths is required")
    if not torch.is_tensor(lengths):
        lengths = torch.as_tensor(lengths)
    if lengths.dtype != torch.int64 or lengths.device != device or not lengths.is_contiguous():
        lengths = lengths.to(device=device, dtype=torch.int64).contiguous()
    if lengths.shape != (B,):
        raise ValueError(f"lengths must have shape ({B},), got {tuple(lengths.shape)}")
    return lengths


def dmv1o_run(dec, attach, lengths, semiring, want_grad, grad_logZ=None, want_dec=True, logZ_shape=None, out=None):
    """Raw launcher.  dec [B,N,2,2,2], attach [B,N,N,2] -> logZ [B] (+ grad_dec, grad_attach fp32).
    want_dec=False: attach counts only (grad_dec is None; the Max semiring then takes the back-pointer walk)."""
    _C.require_gpu(dec, "dmv1o")
    if dec.dim() != 5 or tuple(dec.shape[2:]) != (2, 2, 2):
        raise ValueError(f"dec must be [B,N,2,2,2], got {tuple(dec.shape)}")
    B, N = dec.shape[:2]
    if tuple(attach.shape) != (B, N, N, 2):
        raise ValueError(f"attach must be [B,N,N,2] = {(B, N, N, 2)}, got {tuple(attach.shape)}")
    if dec.dtype != attach.dtype:
        attach = attach.to(dec.dtype)
    dt, dec_c = _C.in_dtype(dec)          # only the storage is read: no detach needed
    _, att_c = _C.in_dtype(attach)
    lengths = _lengths(lengths, B, dec.device)
    L = _C.lib()
    if out is not None:   # caller-owned outputs (allocated on the caller's stream before it switched to a side stream)
        logZ, gdec, gatt = out
    else:
        logZ = torch.empty(logZ_shape or B, dtype=torch.float32, device=dec.device)
    if want_grad:
        if out is None:
            gdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dec.device) if want_dec else None
            gatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dec.device)
        ws, nb = _workspace(_C.OP_DMV1O_INSIDE_OUTSIDE, B, N, semiring, dec.device)
        g = None if grad_logZ is None else grad_logZ.detach().to(torch.float32).reshape(B).contiguous()
        _C.check(L.vlg_dmv1o_inside_outside(_C.ptr(dec_c), _C.ptr(att_c), _C.ptr(lengths), B, N, dt, semiring,
                                            _C.ptr(g), _C.ptr(logZ), _C.ptr(gdec), _C.ptr(gatt), _C.ptr(ws), nb,
                                            _C.stream_of(dec)), "dmv1o_inside_outside")
        return logZ, gdec, gatt
    ws, nb = _workspace(_C.OP_DMV1O_INSIDE, B, N, semiring, dec.device)
    _C.check(L.vlg_dmv1o_inside(_C.ptr(dec_c), _C.ptr(att_c), _C.ptr(lengths), B, N, dt, semiring, _C.ptr(logZ),
                                _C.ptr(ws), nb, _C.stream_of(dec)), "dmv1o_inside")
    return logZ, None, None


def deptree_run(arc, lengths, semiring, want_grad, grad_logZ=None):
    """Raw launcher.  arc [B,N,N] -> logZ [B] (+ grad_arc fp32)."""
    _C.require_gpu(arc, "deptree")
    if arc.dim() != 3:
        raise ValueError("potentials must have dim of 3 (unlabeled)")   # deptree.py:30-31 (labeled: out of scope)
    B, N, N2 = arc.shape
    assert N == N2, "Non-square potentials"                              # deptree.py:149
    dt, arc_c = _C.in_dtype(arc)
    lengths = _lengths(lengths, B, arc.device, allow_none=True)
    logZ = torch.empty(B, dtype=torch.float32, device=arc.device)
    L = _C.lib()
    if want_grad:
        garc = torch.empty((B, N, N), dtype=torch.float32, device=arc.device)
        ws, nb = _workspace(_C.OP_DEPTREE_INSIDE_OUTSIDE, B, N, semiring, arc.device)
        g = None if grad_logZ is None else grad_logZ.detach().to(torch.float32).reshape(B).contiguous()
        _C.check(L.vlg_deptree_inside_outside(_C.ptr(arc_c), _C.ptr(lengths), B, N, dt, semiring, _C.ptr(g),
                                              _C.ptr(logZ), _C.ptr(garc), _C.ptr(ws), nb, _C.stream_of(arc)),
                 "deptree_inside_outside")
        return logZ, garc
    ws, nb = _workspace(_C.OP_DEPTREE_INSIDE, B, N, semiring, arc.device)
    _C.check(L.vlg_deptree_inside(_C.ptr(arc_c), _C.ptr(lengths), B, N, dt, semiring, _C.ptr(logZ), _C.ptr(ws), nb,
                                  _C.stream_of(arc)), "deptree_inside")
    return logZ, None


def dmv1o_decode(dec, attach, lengths, out=None):
    """Viterbi tree as a head vector, entirely on the device (no `nonzero()` host sync).
    Returns (best_score [B], heads [B,N] int64): heads[b,c] = head of word c (0 = root token); 0 at c = 0 / padding.
    Equals `predicted` of src/model/joint.py:256-258 and, shifted by one, of ldndmv.py:301-303."""
    _C.require_gpu(dec, "dmv1o_decode")
    B, N = dec.shape[:2]
    if tuple(dec.shape) != (B, N, 2, 2, 2) or tuple(attach.shape) != (B, N, N, 2):
        raise ValueError(f"dec {tuple(dec.shape)} / attach {tuple(attach.shape)}: expected [B,N,2,2,2] / [B,N,N,2]")
    if dec.dtype != attach.dtype:
        attach = attach.to(dec.dtype)
    dt, dec_c = _C.in_dtype(dec)
    _, att_c = _C.in_dtype(attach)
    lengths = _lengths(lengths, B, dec.device)
    if out is None:
        best = torch.empty(B, dtype=torch.float32, device=dec.device)
        heads = torch.empty((B, N), dtype=torch.int64, device=dec.device)
    else:   # caller-owned outputs (dmv1o_marginals_and_heads allocates them on ITS stream before switching to the side stream)
        best, heads = out
    ws, nb = _workspace(_C.OP_DMV1O_INSIDE_OUTSIDE, B, N, _C.SEMIRING_MAX, dec.device)
    _C.check(_C.lib().vlg_dmv1o_decode(_C.ptr(dec_c), _C.ptr(att_c), _C.ptr(lengths), B, N, dt, _C.ptr(best),
                                       _C.ptr(heads), _C.ptr(ws), nb, _C.stream_of(dec)), "dmv1o_decode")
    return best, heads


def dmv1o_viterbi(dec, attach, lengths, out=None):
    """The Max semiring's every output in one launch: (best [B], grad_dec [B,N,2,2,2], grad_attach [B,N,N,2], heads [B,N]).
    The counts are the 0/1 indicators of the best tree (unit upstream gradient) -- what `-DMV1o(...).max.sum()`
    back-propagates (ldndmv.py:277-281) -- and heads is `argmax` as a head vector (joint.py:256-258)."""
    _C.require_gpu(dec, "dmv1o_viterbi")
    B, N = dec.shape[:2]
    if tuple(dec.shape) != (B, N, 2, 2, 2) or tuple(attach.shape) != (B, N, N, 2):
        raise ValueError(f"dec {tuple(dec.shape)} / attach {tuple(attach.shape)}: expected [B,N,2,2,2] / [B,N,N,2]")
    if dec.dtype != attach.dtype:
        attach = attach.to(dec.dtype)
    dt, dec_c = _C.in_dtype(dec)
    _, att_c = _C.in_dtype(attach)
    lengths = _lengths(lengths, B, dec.device)
    if out is None:
        best = torch.empty(B, dtype=torch.float32, device=dec.device)
        gdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dec.device)
        gatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dec.device)
        heads = torch.empty((B, N), dtype=torch.int64, device=dec.device)
    else:
        best, gdec, gatt, heads = out
    ws, nb = _workspace(_C.OP_DMV1O_INSIDE_OUTSIDE, B, N, _C.SEMIRING_MAX, dec.device)
    _C.check(_C.lib().vlg_dmv1o_viterbi(_C.ptr(dec_c), _C.ptr(att_c), _C.ptr(lengths), B, N, dt, None, _C.ptr(best), _C.ptr(gdec),
                                        _C.ptr(gatt), _C.ptr(heads), _C.ptr(ws), nb, _C.stream_of(dec)), "dmv1o_viterbi")
    return best, gdec, gatt, heads


# ---- one Viterbi pass per training step ------------------------------------------------------------------------------------
# lang_feat_max_tree takes `argmax` of the step's potentials (joint.py:256) and the parser's loss takes `-max` of the SAME
# values a few lines later (ldndmv.py:277-281, through a different DMV1o object built on `.detach()`ed aliases).  When the
# first caller asks for it (`marginals_and_heads(keep_viterbi=True)`), the full Viterbi result is remembered here, keyed by
# the identity of the storage it was computed from: the key tensors are kept alive (so their addresses cannot be handed to
# other data) and their version counters are compared (so an in-place update through any alias invalidates the entry; only
# torch operations count -- a raw pointer write by foreign code is invisible, which is why nothing is remembered unasked).
_VITERBI = {}


def _viterbi_key(dec, attach, lengths):
    return (dec.data_ptr(), attach.data_ptr(), lengths.data_ptr(), dec._version, attach._version, lengths._version,
            tuple(dec.shape), dec.dtype, tuple(dec.stride()), tuple(attach.stride()))


def _viterbi_remember(dec, attach, lengths, result, pending=None):
    """pending: the StructureHandle whose side stream is still producing `result` -- a hit joins it into the CURRENT stream
    before the tensors are handed out, so a `.max` taken before (or on another stream than) `handle.wait()` is ordered too."""
    _VITERBI[dec.device] = (_viterbi_key(dec, attach, lengths), (dec.detach(), attach.detach(), lengths), result, pending)


def _viterbi_lookup(dec, attach, lengths):
    if not isinstance(lengths, torch.Tensor):
        return None
    hit = _VITERBI.get(dec.device)
    if hit is None or hit[0] != _viterbi_key(dec, attach, lengths):
        return None
    if hit[3] is not None:
        hit[3].join_current()
    return hit[2]


def viterbi_forget():
    """Drop the remembered Viterbi result (and the references that keep its potentials alive)."""
    _VITERBI.clear()


def deptree_decode(arc, lengths=None):
    """Best projective single-root tree of arc scores [B,N,N] as heads [B,N] (see dmv1o_decode).  With `arc` = arc
    marginals this is the MBR decode of src/model/ldndmv.py:294-299."""
    _C.require_gpu(arc, "deptree_decode")
    B, N, N2 = arc.shape
    assert N == N2, "Non-square potentials"
    dt, arc_c = _C.in_dtype(arc)
    lengths = _lengths(lengths, B, arc.device, allow_none=True)
    best = torch.empty(B, dtype=torch.float32, device=arc.device)
    heads = torch.empty((B, N), dtype=torch.int64, device=arc.device)
    ws, nb = _workspace(_C.OP_DEPTREE_INSIDE_OUTSIDE, B, N, _C.SEMIRING_MAX, arc.device)
    _C.check(_C.lib().vlg_deptree_decode(_C.ptr(arc_c), _C.ptr(lengths), B, N, dt, _C.ptr(best), _C.ptr(heads),
                                         _C.ptr(ws), nb, _C.stream_of(arc)), "deptree_decode")
    return best, heads


def dmv1o_rules_run(attach_rule, dec, root_rule, token, lengths, semiring, want_grad, head_mask=None, want_heads=False,
                    mask_fill=-1e20, grad_logZ=None):
    """Raw launcher of the rule-table DP (include/vlgae_amd.h: vlg_dmv1o_rules).
    attach_rule [B,L,T,2,2], dec [B,L,2,2,2], root_rule [T] / [1,T] / [B,T], token [B,L] int64, head_mask [B,L] bool.
    Returns dict(logZ [B], grad_rule, grad_dec, grad_root [B,T], heads [B,L+1]) with the requested entries."""
    _C.require_gpu(attach_rule, "dmv1o_rules")
    B, L, T = attach_rule.shape[:3]
    if tuple(attach_rule.shape) != (B, L, T, 2, 2) or tuple(dec.shape) != (B, L, 2, 2, 2) or tuple(token.shape) != (B, L):
        raise ValueError(f"attach_rule {tuple(attach_rule.shape)}, dec {tuple(dec.shape)}, token {tuple(token.shape)}")
    root2 = root_rule.reshape(-1, T)
    if root2.shape[0] not in (1, B):
        raise ValueError(f"root_rule must be [T], [1,T] or [B,T]; got {tuple(root_rule.shape)}")
    dt, rule_c = _C.in_dtype(attach_rule)
    dec_c = dec.detach().to(rule_c.dtype).contiguous()
    root_c = root2.detach().to(rule_c.dtype).contiguous()
    dev = attach_rule.device
    token = token.to(device=dev, dtype=torch.int64).contiguous()
    hm = _C.mask_u8(head_mask, dev)
    lengths = _lengths(lengths, B, dev)
    out = {"logZ": torch.empty(B, dtype=torch.float32, device=dev)}
    g_rule = g_dec = g_root = heads = None
    if want_grad:
        g_rule = torch.empty((B, L, T, 2, 2), dtype=torch.float32, device=dev)
        g_dec = torch.empty((B, L, 2, 2, 2), dtype=torch.float32, device=dev)
        g_root = torch.empty((B, T), dtype=torch.float32, device=dev)
        out.update(grad_rule=g_rule, grad_dec=g_dec, grad_root=g_root)
    if want_heads:
        heads = torch.empty((B, L + 1), dtype=torch.int64, device=dev)
        out["heads"] = heads
    op = _C.OP_DMV1O_INSIDE_OUTSIDE if (want_grad or want_heads) else _C.OP_DMV1O_INSIDE
    ws, nb = _workspace(op, B, L + 1, semiring, dev)
    g = None if grad_logZ is None else grad_logZ.detach().to(torch.float32).reshape(B).contiguous()
    _C.check(_C.lib().vlg_dmv1o_rules(_C.ptr(rule_c), _C.ptr(dec_c), _C.ptr(root_c), int(root2.shape[0] == B and B > 1),
                                      _C.ptr(token), _C.ptr(hm),
                                      _C.ptr(lengths), B, L, T, dt, semiring, float(mask_fill), _C.ptr(g),
                                      _C.ptr(out["logZ"]), _C.ptr(g_rule), _C.ptr(g_dec), _C.ptr(g_root), _C.ptr(heads),
                                      _C.ptr(ws), nb, _C.stream_of(attach_rule)), "dmv1o_rules")
    return out


class _DMV1oRulesSum(torch.autograd.Function):
    """semiring-sum over trees as a function of the scorer's rule tables; backward = rule-space expected counts."""

    @staticmethod
    def forward(ctx, attach_rule, dec, root_rule, token, lengths, head_mask, semiring, mask_fill):
        want = any(ctx.needs_input_grad[:3])
        r = dmv1o_rules_run(attach_rule, dec, root_rule, token, lengths, semiring, want, head_mask, False, mask_fill)
        if want:
            ctx.save_for_backward(r["grad_rule"], r["grad_dec"], r["grad_root"])
        ctx.meta = (attach_rule.dtype, dec.dtype, root_rule.dtype, tuple(root_rule.shape))
        return r["logZ"].unsqueeze(-1)

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        g_rule, g_dec, g_root = ctx.saved_tensors
        d0, d1, d2, root_shape = ctx.meta
        g = grad_out.reshape(-1).to(torch.float32)
        gr = (g_rule * g.view(-1, 1, 1, 1, 1)).to(d0) if ctx.needs_input_grad[0] else None
        gd = (g_dec * g.view(-1, 1, 1, 1, 1)).to(d1) if ctx.needs_input_grad[1] else None
        gt = None
        if ctx.needs_input_grad[2]:
            gt = g_root * g.view(-1, 1)
            B, T = gt.shape
            per_sentence = len(root_shape) == 2 and root_shape[0] == B and B > 1
            gt = (gt if per_sentence else gt.sum(0)).reshape(root_shape).to(d2)   # a shared root table sums over the batch
        return gr, gd, gt, None, None, None, None, None


def dmv1o_rules_sum(attach_rule, dec, root_rule, token, lengths, head_mask=None, semiring=0, mask_fill=-1e20):
    return _DMV1oRulesSum.apply(attach_rule, dec, root_rule, token, lengths, head_mask, semiring, mask_fill)


class _DMV1oSum(torch.autograd.Function):
    """semiring-sum over all trees; d/d(potentials) = expected counts (Log) / best tree (Max)."""

    @staticmethod
    def forward(ctx, dec, attach, lengths, semiring):
        want = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        hit = _viterbi_lookup(dec, attach, lengths) if semiring == _C.SEMIRING_MAX and dec.dtype != torch.float64 else None
        if hit is not None:   # this step's Viterbi pass already ran on these very values (marginals_and_heads(keep_viterbi=True))
            best, gdec, gatt, _ = hit
            logZ = best.view(-1, 1)
        else:
            logZ, gdec, gatt = dmv1o_run(dec, attach, lengths, semiring, want, logZ_shape=(dec.shape[0], 1))   # [B,1], helpers.py:116
        if want:
            ctx.save_for_backward(gdec, gatt)
            ctx.fwd_thread = threading.get_ident()
        ctx.in_dtypes = (dec.dtype, attach.dtype)
        return logZ.double() if dec.dtype == torch.float64 else logZ

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        _note_backward_thread(ctx)
        gdec, gatt = ctx.saved_tensors
        want_d, want_a = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gd, ga = _scale_counts(gdec if want_d else None, gatt if want_a else None, grad_out, ctx.in_dtypes[0], ctx.in_dtypes[1])
        return gd, ga, None, None


_WARNED_ENGINE_THREAD = False


def _note_backward_thread(ctx):
    """Warn once when backward runs on torch's autograd engine thread instead of the thread that ran forward."""
    global _WARNED_ENGINE_THREAD
    if not _WARNED_ENGINE_THREAD and getattr(ctx, "fwd_thread", None) not in (None, threading.get_ident()):
        _WARNED_ENGINE_THREAD = True
        warnings.warn("vlgae_amd: backward is running on torch's autograd engine thread; the hand-off costs 50-120 us per call, "
                      "more than the fused DP kernel takes. A one-process-per-GPU trainer should call "
                      "vlgae_amd.configure_autograd() (= torch.autograd.set_multithreading_enabled(False)) once at start-up.",
                      RuntimeWarning, stacklevel=2)


def _scale_counts(ca, cb, grad_out, dtype_a, dtype_b):
    """counts * d loss / d logZ, cast to the potentials' dtypes: one launch (vlg_scale_counts) for the fp32 / bf16 cases."""
    first = ca if ca is not None else cb
    if first is None:
        return None, None
    B = first.shape[0]
    dt = dtype_a if ca is not None else dtype_b
    if (ca is not None and cb is not None and dtype_a != dtype_b) or (dt != torch.float32 and dt != torch.bfloat16) or B == 0:
        g = grad_out.reshape(-1).to(torch.float32)
        return tuple(None if c is None else (c * g.view(-1, *([1] * (c.dim() - 1)))).to(d) for c, d in ((ca, dtype_a), (cb, dtype_b)))
    if grad_out.dtype != torch.float32:
        grad_out = grad_out.float()
    stride = 1
    if grad_out.stride() == (0,) * grad_out.dim():
        stride = 0                                                   # the expanded scalar a `.sum()` hands back
    elif not grad_out.is_contiguous():
        grad_out = grad_out.contiguous()
    oa = None if ca is None else torch.empty(ca.shape, dtype=dt, device=ca.device)
    ob = None if cb is None else torch.empty(cb.shape, dtype=dt, device=cb.device)
    _C.check(_C.lib().vlg_scale_counts(_C.ptr(ca), _C.ptr(cb), _C.ptr(grad_out), stride, B, 0 if ca is None else ca.numel() // B,
                                       0 if cb is None else cb.numel() // B, _C.BF16 if dt == torch.bfloat16 else _C.F32, _C.ptr(oa),
                                       _C.ptr(ob), _C.stream_of(first)), "scale_counts")
    return oa, ob


class _DepTreeSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, arc, lengths, semiring):
        want = ctx.needs_input_grad[0]
        logZ, garc = deptree_run(arc, lengths, semiring, want)
        if want:
            ctx.save_for_backward(garc)
        ctx.in_dtype = arc.dtype
        return logZ.to(arc.dtype if arc.dtype == torch.float64 else torch.float32)                  # [B], deptree.py:75

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        (garc,) = ctx.saved_tensors
        return _scale_counts(garc, None, grad_out, ctx.in_dtype, ctx.in_dtype)[0], None, None


def dmv1o_sum(dec, attach, lengths, semiring):
    return _DMV1oSum.apply(dec, attach, lengths, semiring)


def deptree_sum(arc, lengths, semiring):
    return _DepTreeSum.apply(arc, lengths, semiring)


def dmv1o_merge(dec, attach, root, one=0.0, zero=NEGINF):
    """DMV1o.merge (distributions.py:253-265): root-augmented potentials, always float32."""
    _C.require_gpu(dec, "dmv1o_merge")
    B, Lw = dec.shape[:2]
    if tuple(dec.shape) != (B, Lw, 2, 2, 2) or tuple(attach.shape) != (B, Lw, Lw, 2) or tuple(root.shape) != (B, Lw):
        raise ValueError(f"merge: dec {tuple(dec.shape)}, attach {tuple(attach.shape)}, root {tuple(root.shape)}")
    dt, dec_c = _C.in_dtype(dec)
    attach = attach.detach().to(dec_c.dtype).contiguous()
    root = root.detach().to(dec_c.dtype).contiguous()
    N = Lw + 1
    dec_w = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dec.device)
    att_w = torch.empty((B, N, N, 2), dtype=torch.float32, device=dec.device)
    _C.check(_C.lib().vlg_dmv1o_merge(_C.ptr(dec_c), _C.ptr(attach), _C.ptr(root), B, Lw, dt, float(one), float(zero),
                                      _C.ptr(dec_w), _C.ptr(att_w), _C.stream_of(dec)), "dmv1o_merge")
    return dec_w, att_w


class _Merge(torch.autograd.Function):
    """merge is fill + copy; its adjoint is three slices (the reference gets this from autograd on the
    in-place index assignments of distributions.py:260-264)."""

    @staticmethod
    def forward(ctx, dec, attach, root, one, zero):
        ctx.in_dtypes = (dec.dtype, attach.dtype, root.dtype)
        return dmv1o_merge(dec, attach, root, one, zero)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_dec_w, g_att_w):
        d0, d1, d2 = ctx.in_dtypes
        return (g_dec_w[:, 1:].to(d0), g_att_w[:, 1:, 1:, :].to(d1), g_att_w[:, 0, 1:, 1].to(d2), None, None)


def dmv1o_merge_autograd(dec, attach, root, one=0.0, zero=NEGINF):
    return _Merge.apply(dec, attach, root, one, zero)


_SIDE_STREAMS = {}


def _marginals_viterbi_one_launch(dec, attach, lengths, keep_viterbi):
    """Both DPs of `dmv1o_marginals_and_heads` as ONE launch on the current stream (vlg_dmv1o_marginals_viterbi: grid (B, 2), the two
    workgroups of a sentence share a CU): sentences short enough that both passes keep their charts in LDS (N <= 44: twice the larger footprint within 160 KB).  No side
    stream, no events: ~15 us less than the two-stream form at B = 256, L = 40 and nothing for a HIP-graph capture to fork."""
    if tuple(dec.shape[2:]) != (2, 2, 2):
        raise ValueError(f"dec must be [B,N,2,2,2], got {tuple(dec.shape)}")
    B, N = dec.shape[:2]
    if tuple(attach.shape) != (B, N, N, 2):
        raise ValueError(f"attach must be [B,N,N,2] = {(B, N, N, 2)}, got {tuple(attach.shape)}")
    if dec.dtype != attach.dtype:
        attach = attach.to(dec.dtype)
    dt, dec_c = _C.in_dtype(dec)
    _, att_c = _C.in_dtype(attach)
    lengths = _lengths(lengths, B, dec.device)
    dev = dec.device
    logZ, best = torch.empty(B, dtype=torch.float32, device=dev), torch.empty(B, dtype=torch.float32, device=dev)
    gatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dev)
    heads = torch.empty((B, N), dtype=torch.int64, device=dev)
    vdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dev) if keep_viterbi else None
    vatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dev) if keep_viterbi else None
    _C.check(_C.lib().vlg_dmv1o_marginals_viterbi(_C.ptr(dec_c), _C.ptr(att_c), _C.ptr(lengths), B, N, dt, _C.ptr(logZ), None, _C.ptr(gatt),
                                                  _C.ptr(best), _C.ptr(vdec), _C.ptr(vatt), _C.ptr(heads), _C.stream_of(dec)),
             "dmv1o_marginals_viterbi")
    if keep_viterbi:
        _viterbi_remember(dec, attach, lengths, (best, vdec, vatt, heads))
    return logZ, gatt, heads


def dmv1o_marginals_and_heads(dec, attach, lengths, keep_viterbi=False):
    """What lang_feat_max_tree needs from one sentence batch (joint.py:251-258): the arc marginals
    d logZ / d attach AND the Viterbi heads.  The two are independent DPs over the same potentials; at one
    workgroup per CU each leaves most of the machine idle and their LDS footprints (78 KB + 49 KB at N = 41) fit one
    CU together, so the decode runs on a side HIP stream next to the inside-outside launch and joins before returning.
    Returns (logZ [B], marginals [B,N,N,2], heads [B,N]).

    keep_viterbi=True (training with `viterbi_training`, ldndmv.py:277-281): the side-stream launch is the full Viterbi pass
    (best score + tree counts + heads, 67 us instead of the 57 us walk) and its result is remembered, so the `DMV1o(...).max`
    that the loss takes of the same potentials later in the step launches nothing."""
    _C.require_gpu(dec, "dmv1o_marginals_and_heads")
    if dec.dim() == 5 and _C.lib().vlg_dmv1o_marginals_viterbi_supported(dec.shape[1]):
        return _marginals_viterbi_one_launch(dec, attach, lengths, keep_viterbi)
    cur = torch.cuda.current_stream(dec.device)
    side = _SIDE_STREAMS.get(dec.device)
    if side is None:
        side = _SIDE_STREAMS[dec.device] = torch.cuda.Stream(device=dec.device)
    # The decode's outputs are allocated HERE, on the current stream, and the side stream joins before this returns: every later
    # use, free or reuse of them -- and of the inputs -- is ordered behind the side stream's work by that join, so no
    # `record_stream` bookkeeping is needed (it is what made a HIP-graph capture crash while tensors of an earlier eager
    # step were still alive: the allocator's deferred event handling for recorded streams is not capture-safe).
    B, N = dec.shape[:2]
    best = torch.empty(B, dtype=torch.float32, device=dec.device)
    heads = torch.empty((B, N), dtype=torch.int64, device=dec.device)
    if keep_viterbi:
        vdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dec.device)
        vatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dec.device)
        lengths = _lengths(lengths, B, dec.device)
    # The potentials are produced on the current stream: the side stream waits for an event recorded HERE, and the longer of the
    # two launches (the Log-semiring inside-outside pass) is enqueued first -- it starts ~6 us earlier than when it followed the
    # side-stream launch, and the pair ends when it does.
    ready = torch.cuda.Event()
    ready.record(cur)
    logZ, _, gatt = dmv1o_run(dec, attach, lengths, _C.SEMIRING_LOG, True, want_dec=False)
    side.wait_event(ready)
    with torch.cuda.stream(side):
        if keep_viterbi:
            dmv1o_viterbi(dec, attach, lengths, out=(best, vdec, vatt, heads))
        else:
            dmv1o_decode(dec, attach, lengths, out=(best, heads))
    cur.wait_stream(side)
    if keep_viterbi:
        _viterbi_remember(dec, attach, lengths, (best, vdec, vatt, heads))
    return logZ, gatt, heads


class StructureHandle:
    """Both DPs of lang_feat_max_tree in flight on two side streams (see dmv1o_structure_async); `wait()` joins them into the
    current stream and returns (logZ [B], marginals [B,N,N,2], heads [B,N])."""

    def __init__(self, streams, outs, device):
        self._streams, self._outs, self._device, self._joined = streams, outs, device, set()

    def join_current(self):
        """Orders the CURRENT stream behind both side streams (idempotent per stream; cheap: two event waits)."""
        cur = torch.cuda.current_stream(self._device)
        if cur not in self._joined:
            for st in self._streams:
                cur.wait_stream(st)
            self._joined.add(cur)

    def wait(self):
        self.join_current()
        return self._outs


_SIDE_STREAMS2 = {}


def dmv1o_structure_async(dec, attach, lengths, keep_viterbi=False):
    """`dmv1o_marginals_and_heads` without occupying the current stream: the Log-semiring inside-outside pass AND the Viterbi
    pass go to two side streams (ordered after what the current stream has enqueued so far: the potentials), and the caller
    keeps enqueueing work that does not need them -- in the training step the attention-fuse, the root row and the encoders'
    projection GEMM (37 us of kernels) run beside the 93 us the DPs take at one workgroup per CU each.  `.wait()` joins.
    Outputs are allocated here, on the current stream (the reasoning of dmv1o_marginals_and_heads applies)."""
    _C.require_gpu(dec, "dmv1o_structure_async")
    dev = dec.device
    cur = torch.cuda.current_stream(dev)
    pair = _SIDE_STREAMS2.get(dev)
    if pair is None:
        pair = _SIDE_STREAMS2[dev] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
    B, N = dec.shape[:2]
    lengths = _lengths(lengths, B, dev)
    logZ = torch.empty(B, dtype=torch.float32, device=dev)
    gatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dev)
    best = torch.empty(B, dtype=torch.float32, device=dev)
    heads = torch.empty((B, N), dtype=torch.int64, device=dev)
    if keep_viterbi:
        vdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dev)
        vatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dev)
    for st in pair:
        st.wait_stream(cur)
    with torch.cuda.stream(pair[0]):
        dmv1o_run(dec, attach, lengths, _C.SEMIRING_LOG, True, want_dec=False, out=(logZ, None, gatt))
    with torch.cuda.stream(pair[1]):
        if keep_viterbi:
            dmv1o_viterbi(dec, attach, lengths, out=(best, vdec, vatt, heads))
        else:
            dmv1o_decode(dec, attach, lengths, out=(best, heads))
    handle = StructureHandle(pair, (logZ, gatt, heads), dev)
    if keep_viterbi:   # the entry carries the handle: a lookup joins the side streams before it returns the hit
        _viterbi_remember(dec, attach, lengths, (best, vdec, vatt, heads), pending=handle)
    return handle
