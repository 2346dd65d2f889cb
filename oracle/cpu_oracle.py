"""ctypes/numpy front-end of oracle/vlg_oracle.c, plus brute-force tree enumerators.

TEST INFRASTRUCTURE.  Parity status: PINNED against outputs of the reference itself
(tests/golden/*.npz via tests/golden/make_golden.py) -- see tests/test_oracle_golden.py.
"""
import ctypes
import itertools
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("VLG_ORACLE_SO") or os.path.join(_HERE, "_build", "libvlg_oracle.so")   # override: a sanitizer build
NEGINF = -1e12  # src/model/torch_struct/semirings/semirings.py:16 (standalone value)
_lib = None


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("vlg_oracle.c", "vlg_oracle_impl.h")]
    stale = (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if (force or stale) and not os.environ.get("VLG_ORACLE_SO"):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.orc_max_threads.restype = ctypes.c_int
    return _lib


def max_threads():
    return int(_load().orc_max_threads())


def set_threads(n):
    _load().orc_set_threads(int(n))


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _suffix(dtype):
    dtype = np.dtype(dtype)
    assert dtype in (np.float32, np.float64)
    return dtype, ("_f32" if dtype == np.float32 else "_f64")


def dmv1o_merge(dec, attach, root, one=0.0, zero=NEGINF):
    """DMV1o.merge, src/model/torch_struct/distributions.py:253-265 (always float32 out)."""
    B, L = dec.shape[:2]
    N = L + 1
    attach_wroot = np.full((B, N, N, 2), zero, dtype=np.float32)
    dec_wroot = np.full((B, N, 2, 2, 2), zero, dtype=np.float32)
    attach_wroot[:, 0, 1:, 1] = root          # NOCHILD = 1
    attach_wroot[:, 1:, 1:, :] = attach
    dec_wroot[:, 0, 1, :, :] = one            # RIGHT = 1
    dec_wroot[:, 1:] = dec
    return dec_wroot, attach_wroot


def dmv1o(dec, attach, lengths, semiring="log", dtype=np.float32, grad=True, glogZ=None, neg_inf=NEGINF):
    """DMV1oStruct._dp (dmv.py:19-66) + its autograd outside.  dec [B,N,2,2,2], attach [B,N,N,2].
    Returns logZ [B,1] (helpers.py:101-116 keeps the trailing 1), grad_dec, grad_attach."""
    dtype, suf = _suffix(dtype)
    dec = np.ascontiguousarray(dec, dtype=dtype)
    attach = np.ascontiguousarray(attach, dtype=dtype)
    lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    B, N = dec.shape[:2]
    assert dec.shape == (B, N, 2, 2, 2) and attach.shape == (B, N, N, 2) and lengths.shape == (B,)
    logZ = np.empty((B,), dtype=dtype)
    gdec = np.empty_like(dec) if grad else None
    gatt = np.empty_like(attach) if grad else None
    if glogZ is not None:
        glogZ = np.ascontiguousarray(glogZ, dtype=dtype).reshape(B)
    rc = getattr(_load(), "orc_dmv1o" + suf)(
        _p(dec), _p(attach), _p(lengths), B, N, 0 if semiring == "log" else 1, ctypes.c_double(neg_inf),
        _p(glogZ), _p(logZ), _p(gdec), _p(gatt))
    assert rc == 0, rc
    return logZ.reshape(B, 1), gdec, gatt


def dmv1o_rules(attach_rule, dec, root_rule, token, lengths, head_mask=None, semiring="log", dtype=np.float32,
                grad=True, mask_fill=-1e20):
    """Scorer -> DP glue of DiscriminativeNDMV._forward restated in numpy (src/model/ldndmv.py:189-209), then the
    DP oracle, then the adjoint of the glue (scatter-add over repeated tokens).
      attach_rule [B,L,T,2(dir),2(val)], dec [B,L,2,2,2], root_rule [T] or [1,T] or [B,T], token [B,L] in [0,T)
    Returns logZ [B,1], grad_rule [B,L,T,2,2], grad_dec [B,L,2,2,2], grad_root [B,T] (per sentence), heads-free."""
    dtype = np.dtype(dtype)
    attach_rule = np.asarray(attach_rule, dtype=dtype)
    dec = np.asarray(dec, dtype=dtype)
    B, L, T = attach_rule.shape[:3]
    root_rule = np.broadcast_to(np.asarray(root_rule, dtype=dtype).reshape(-1, T), (B, T))
    token = np.asarray(token, dtype=np.int64)
    bi = np.arange(B)[:, None, None]
    hi = np.arange(L)[None, :, None]
    ci = np.arange(L)[None, None, :]
    tok = token[:, None, :]                                              # child's token id
    left = attach_rule[bi, hi, tok, 0, :]                                # [B,L(head),L(child),val]  (:189-190)
    right = attach_rule[bi, hi, tok, 1, :]
    attach = np.where((ci < hi)[..., None], left, np.where((ci > hi)[..., None], right, 0))   # tril / triu  (:191-194)
    if head_mask is not None:
        attach = np.where(np.asarray(head_mask, dtype=bool)[:, :, None, None], dtype.type(mask_fill), attach)   # :195-199
    root = np.take_along_axis(root_rule, token, axis=1)                 # :207
    N = L + 1                                                           # DMV1o.merge, distributions.py:253-265
    ma = np.full((B, N, N, 2), NEGINF, dtype=dtype)
    md = np.full((B, N, 2, 2, 2), NEGINF, dtype=dtype)
    ma[:, 0, 1:, 1] = root
    ma[:, 1:, 1:, :] = attach
    md[:, 0, 1, :, :] = 0
    md[:, 1:] = dec
    logZ, gmd, gma = dmv1o(md, ma, lengths, semiring, dtype, grad=grad)
    if not grad:
        return logZ, None, None, None
    g_att = gma[:, 1:, 1:, :].copy()
    if head_mask is not None:
        g_att[np.asarray(head_mask, dtype=bool)] = 0                   # masked_fill: no gradient
    g_rule = np.zeros_like(attach_rule)
    bb, hh, cc = np.broadcast_arrays(bi, hi, ci)
    tt = np.broadcast_to(tok, (B, L, L))
    lm, rm = (cc < hh), (cc > hh)
    for v in range(2):
        np.add.at(g_rule, (bb[lm], hh[lm], tt[lm], 0, v), g_att[..., v][lm])
        np.add.at(g_rule, (bb[rm], hh[rm], tt[rm], 1, v), g_att[..., v][rm])
    g_root = np.zeros((B, T), dtype=dtype)
    np.add.at(g_root, (np.arange(B)[:, None].repeat(L, 1), token), gma[:, 0, 1:, 1])
    return logZ, g_rule, gmd[:, 1:].copy(), g_root


def deptree(arc, lengths=None, semiring="log", dtype=np.float32, grad=True, glogZ=None, neg_inf=NEGINF):
    """DepTree._dp (deptree.py:25-76) + its autograd outside.  arc [B,N,N] head->child, root = 0.
    Returns logZ [B], grad_arc [B,N,N]."""
    dtype, suf = _suffix(dtype)
    arc = np.ascontiguousarray(arc, dtype=dtype)
    B, N = arc.shape[:2]
    assert arc.shape == (B, N, N), "Non-square potentials"
    if lengths is not None:
        lengths = np.ascontiguousarray(lengths, dtype=np.int64)
    logZ = np.empty((B,), dtype=dtype)
    garc = np.empty_like(arc) if grad else None
    if glogZ is not None:
        glogZ = np.ascontiguousarray(glogZ, dtype=dtype).reshape(B)
    rc = getattr(_load(), "orc_deptree" + suf)(
        _p(arc), _p(lengths), B, N, 0 if semiring == "log" else 1, ctypes.c_double(neg_inf), _p(glogZ), _p(logZ),
        _p(garc))
    assert rc == 0, rc
    return logZ, garc


def bilinear_align(txt, vis, tmask=None, vmask=None, dtype=np.float32, neg_inf=-1e20, full=True, maxV=False,
                   maxQ=False, diag=False):
    """gather_logit_simple, src/model/joint.py:406-419.  txt [B,Q,d], vis [A,V,d] -> attmap [B,A,Q,V]."""
    dtype, suf = _suffix(dtype)
    txt = np.ascontiguousarray(txt, dtype=dtype)
    vis = np.ascontiguousarray(vis, dtype=dtype)
    B, Q, d = txt.shape
    A, V, d2 = vis.shape
    assert d == d2
    tm = None if tmask is None else np.ascontiguousarray(tmask, dtype=np.uint8)
    vm = None if vmask is None else np.ascontiguousarray(vmask, dtype=np.uint8)
    o_full = np.empty((B, A, Q, V), dtype=dtype) if full else None
    o_maxV = np.empty((B, A, Q), dtype=dtype) if maxV else None
    o_maxQ = np.empty((B, A, V), dtype=dtype) if maxQ else None
    o_diag = np.empty((B, Q, V), dtype=dtype) if diag else None
    rc = getattr(_load(), "orc_bilinear_align" + suf)(
        _p(txt), _p(vis), _p(tm), _p(vm), B, A, Q, V, d, ctypes.c_double(neg_inf), _p(o_full), _p(o_maxV), _p(o_maxQ),
        _p(o_diag))
    assert rc == 0, rc
    return dict(full=o_full, maxV=o_maxV, maxQ=o_maxQ, diag=o_diag)


def bilinear_align_backward(g, txt, vis, tmask=None, vmask=None, dtype=np.float64):
    """Adjoint of `bilinear_align` (joint.py:413-418 under autograd) for a cotangent g [B,A,Q,V]: (g_txt [B,Q,d], g_vis [A,V,d])."""
    dtype, suf = _suffix(dtype)
    g, txt, vis = (np.ascontiguousarray(x, dtype=dtype) for x in (g, txt, vis))
    B, Q, d = txt.shape
    A, V, _ = vis.shape
    assert g.shape == (B, A, Q, V)
    tm = None if tmask is None else np.ascontiguousarray(tmask, dtype=np.uint8)
    vm = None if vmask is None else np.ascontiguousarray(vmask, dtype=np.uint8)
    g_txt, g_vis = np.empty_like(txt), np.empty_like(vis)
    rc = getattr(_load(), "orc_bilinear_align_bwd" + suf)(_p(g), _p(txt), _p(vis), _p(tm), _p(vm), B, A, Q, V, d, _p(g_txt), _p(g_vis))
    assert rc == 0, rc
    return g_txt, g_vis


def attn_fuse(vis, txt, vis_mid, enc_x, gamma, beta, eps=1e-5, dtype=np.float32):
    """Attention-fuse, src/model/joint.py:670-674.  Returns (attmap [B,L,V], out [B,L,h])."""
    dtype, suf = _suffix(dtype)
    vis, txt, vis_mid, enc_x, gamma, beta = (np.ascontiguousarray(x, dtype=dtype)
                                             for x in (vis, txt, vis_mid, enc_x, gamma, beta))
    B, V, d = vis.shape
    L = txt.shape[1] - 1
    h = vis_mid.shape[2]
    assert txt.shape == (B, L + 1, d) and vis_mid.shape == (B, V, h) and enc_x.shape == (B, L, h)
    att = np.empty((B, L, V), dtype=dtype)
    out = np.empty((B, L, h), dtype=dtype)
    rc = getattr(_load(), "orc_attn_fuse" + suf)(
        _p(vis), _p(txt), _p(vis_mid), _p(enc_x), _p(gamma), _p(beta), B, L, V, d, h, ctypes.c_double(eps), _p(att),
        _p(out))
    assert rc == 0, rc
    return att, out


def attn_fuse_backward(vis, txt, vis_mid, enc_x, gamma, dout, eps=1e-5, dtype=np.float32):
    """Adjoint of `attn_fuse` for the cotangent `dout` [B,L,h] of its output.
    Returns (d_vis, d_txt, d_vis_mid, d_enc_x, d_gamma, d_beta); d_txt[:, 0] (the root slot) is zero."""
    dtype, suf = _suffix(dtype)
    vis, txt, vis_mid, enc_x, gamma, dout = (np.ascontiguousarray(x, dtype=dtype)
                                             for x in (vis, txt, vis_mid, enc_x, gamma, dout))
    B, V, d = vis.shape
    L = txt.shape[1] - 1
    h = vis_mid.shape[2]
    assert txt.shape == (B, L + 1, d) and vis_mid.shape == (B, V, h) and enc_x.shape == (B, L, h) and dout.shape == (B, L, h)
    outs = [np.empty_like(vis), np.empty_like(txt), np.empty_like(vis_mid), np.empty_like(enc_x), np.empty(h, dtype=dtype),
            np.empty(h, dtype=dtype)]
    rc = getattr(_load(), "orc_attn_fuse_bwd" + suf)(
        _p(vis), _p(txt), _p(vis_mid), _p(enc_x), _p(gamma), _p(dout), B, L, V, d, h, ctypes.c_double(eps),
        *(_p(o) for o in outs))
    assert rc == 0, rc
    return tuple(outs)


def arc_encoder(child, parent, w1, w2=None, b=None, dtype=np.float32):
    """joint.py:281-287: einsum('bcx,xhy,bcy->bch', child, w1, parent) + (child + parent) @ w2 + b.  [..., X] inputs."""
    dtype, suf = _suffix(dtype)
    lead = child.shape[:-1]
    child, parent, w1 = (np.ascontiguousarray(a, dtype=dtype) for a in (child, parent, w1))
    X, H, Y = w1.shape
    M = int(np.prod(lead))
    assert child.shape[-1] == X and parent.shape == lead + (Y,) and (w2 is None or X == Y)
    w2c = None if w2 is None else np.ascontiguousarray(w2, dtype=dtype)
    bc = None if b is None else np.ascontiguousarray(b, dtype=dtype)
    out = np.empty(lead + (H,), dtype=dtype)
    rc = getattr(_load(), "orc_arc_encoder" + suf)(_p(child), _p(parent), _p(w1), _p(w2c), _p(bc), M, X, H, Y, _p(out))
    assert rc == 0, rc
    return out


def arc_encoder_backward(child, parent, w1, w2, g, dtype=np.float32):
    """Adjoint of `arc_encoder` for the cotangent g [..., H]: (d_child, d_parent, d_w1, d_w2 or None, d_b)."""
    dtype, suf = _suffix(dtype)
    lead = child.shape[:-1]
    child, parent, w1, g = (np.ascontiguousarray(a, dtype=dtype) for a in (child, parent, w1, g))
    X, H, Y = w1.shape
    M = int(np.prod(lead))
    w2c = None if w2 is None else np.ascontiguousarray(w2, dtype=dtype)
    d_child, d_parent, d_w1 = np.empty_like(child), np.empty_like(parent), np.empty_like(w1)
    d_w2 = None if w2 is None else np.empty_like(w2c)
    d_b = np.empty(H, dtype=dtype)
    rc = getattr(_load(), "orc_arc_encoder_bwd" + suf)(_p(child), _p(parent), _p(w1), _p(w2c), _p(g), M, X, H, Y, _p(d_child),
                                                       _p(d_parent), _p(d_w1), _p(d_w2), _p(d_b))
    assert rc == 0, rc
    return d_child, d_parent, d_w1, d_w2, d_b


def grounding_prior(tag, factor_names, vis_split, pos_for, Q):
    """The additive POS prior of joint.py:446-470 as a table: returns (pen [B,Q,S] float64, seg_of_v [V] uint8).
    For every named factor f in {obj, rel, attr} whose POS set contains the token's tag, every region OUTSIDE f's
    segment loses 100; the rows touched are q = 1..L (the word queries), the root slot and the arc queries keep 0."""
    tag = np.asarray(tag)
    B, L = tag.shape
    S = len(vis_split)
    seg_of_v = np.repeat(np.arange(S, dtype=np.uint8), np.asarray(vis_split, dtype=np.int64))
    pen = np.zeros((B, Q, S), dtype=np.float64)
    for f, name in enumerate(factor_names):
        if str(name) not in ("obj", "rel", "attr"):
            continue
        hit = np.isin(tag, np.asarray(pos_for[str(name)]))          # [B,L]
        for s_ in range(S):
            if s_ != f:
                pen[:, 1:L + 1, s_] += 100.0 * hit
    return pen, seg_of_v


def grounding_loss(txt, vis, tmask, vmask, marginal, num_token, w_vis2txt=1.0, pen=None, seg_of_v=None,
                   neg_inf=-1e20, dtype=np.float32, want_grad=True):
    """gather_logit_simple + loss_grounding_factor_ce (joint.py:406-419, 439-491), A == B.
    Returns dict(txt2vis, vis2txt, total, maxV, argV, maxQ, argQ[, g_txt, g_vis])."""
    dtype, suf = _suffix(dtype)
    txt, vis, marginal = (np.ascontiguousarray(x, dtype=dtype) for x in (txt, vis, marginal))
    B, Q, d = txt.shape
    A, V, d2 = vis.shape
    assert A == B and d == d2 and marginal.shape == (B, Q)
    tm = None if tmask is None else np.ascontiguousarray(tmask, dtype=np.uint8)
    vm = None if vmask is None else np.ascontiguousarray(vmask, dtype=np.uint8)
    n_seg = 0
    if pen is not None:
        pen = np.ascontiguousarray(pen, dtype=dtype)
        seg_of_v = np.ascontiguousarray(seg_of_v, dtype=np.uint8)
        n_seg = pen.shape[2]
        assert pen.shape == (B, Q, n_seg) and seg_of_v.shape == (V,)
    sums = np.empty(3, dtype=dtype)
    maxV, argV = np.empty((B, A, Q), dtype=dtype), np.empty((B, A, Q), dtype=np.int32)
    maxQ, argQ = np.empty((B, A, V), dtype=dtype), np.empty((B, A, V), dtype=np.int32)
    g_txt = np.empty_like(txt) if want_grad else None
    g_vis = np.empty_like(vis) if want_grad else None
    rc = getattr(_load(), "orc_grounding_loss" + suf)(
        _p(txt), _p(vis), _p(tm), _p(vm), _p(marginal), _p(pen), _p(seg_of_v), n_seg, B, Q, V, d, ctypes.c_double(neg_inf),
        ctypes.c_double(float(num_token)), ctypes.c_double(float(w_vis2txt)), _p(sums), _p(maxV), _p(argV), _p(maxQ),
        _p(argQ), _p(g_txt), _p(g_vis))
    assert rc == 0, rc
    out = dict(txt2vis=float(sums[0]), vis2txt=float(sums[1]), total=float(sums[2]), maxV=maxV, argV=argV, maxQ=maxQ, argQ=argQ)
    if want_grad:
        out.update(g_txt=g_txt, g_vis=g_vis)
    return out


# ----------------------------------------------------------------------------------------------
# Brute-force enumerators (pure Python, tiny N only): an algorithm-independent known answer.
# ----------------------------------------------------------------------------------------------
def gather_logit_reduced(txt, vis, tmask, vmask, marginal, g_logit=None, neg_inf=-1e20, dtype=np.float64):
    """gather_logit_simple + gather_logit_reduced (joint.py:406-432): logit[b,a] = sum_q marginal[b,q] max_v att[b,a,q,v] /
    sum_q marginal[b,q].  With g_logit [B,A] also the gradients to txt and vis: through the first arg-max of every maximum
    and only where both masks are on (masked_fill_ passes no gradient)."""
    al = bilinear_align(txt, vis, tmask, vmask, dtype, neg_inf, full=True)
    att = al["full"].astype(np.float64)
    marginal = np.asarray(marginal, dtype=np.float64)
    den = marginal.sum(1, keepdims=True)                                        # [B,1]
    logit = (att.max(-1) * marginal[:, None, :]).sum(-1) / den
    out = dict(logit=logit)
    if g_logit is not None:
        B, A, Q, V = att.shape
        arg = att.argmax(-1)                                                    # [B,A,Q] first maximum
        tm = np.ones((B, Q), bool) if tmask is None else np.asarray(tmask, bool)
        vm = np.ones((A, V), bool) if vmask is None else np.asarray(vmask, bool)
        coef = np.asarray(g_logit, np.float64)[:, :, None] * (marginal / den)[:, None, :]      # [B,A,Q]
        coef = coef * tm[:, None, :] * vm[np.arange(A)[None, :, None], arg]
        t64, v64 = np.asarray(txt, np.float64), np.asarray(vis, np.float64)
        g_txt = np.einsum("baq,baqd->bqd", coef, v64[np.arange(A)[None, :, None], arg])
        g_vis = np.zeros_like(v64)
        for b in range(B):
            for a in range(A):
                np.add.at(g_vis[a], arg[b, a], coef[b, a][:, None] * t64[b])
        out.update(g_txt=g_txt, g_vis=g_vis)
    return out


def grounding_decode(diag, max_v, tag, factor_names, vis_split, pos_for, use_pos_prior=True, use_heuristic=True):
    """Tensor half of decode_grounding_on_factor (joint.py:516-596) in float32, the reference's dtype.
    diag [B,Q,V] = batch diagonal of the alignment tensor (:522-525), max_v [B,A,Q] = its max over V; tag [B,L].
    Returns dict(logit = the edited diagonal block, top5 [B,Q,min(5,V)] (descending; equal values by ascending column),
    factor2img [B,Q])."""
    x = np.array(diag, dtype=np.float32, copy=True)
    tag = np.asarray(tag)
    B, Q, V = x.shape
    L = tag.shape[1]
    names = [str(n) for n in factor_names]
    widths = [int(w) for w in vis_split]
    factor2img = np.argmax(np.asarray(max_v), axis=1)                    # :520 (first maximum)
    if use_pos_prior:                                                     # :528-552
        offset = 0
        for name, width in zip(names, widths):
            if name in ("obj", "rel", "attr"):
                hit = np.isin(tag, np.asarray(pos_for[name])).astype(np.float32)[..., None]   # [B,L,1]
                x[:, 1:L + 1, :offset] -= np.float32(1e10) * hit
                x[:, 1:L + 1, offset + width:] -= np.float32(1e10) * hit
            offset += width
    if use_heuristic:                                                     # :554-594
        n_box = widths[0]
        starts = np.concatenate([[0], np.cumsum(widths)])
        aligned = x.max(-1)                                               # [B,Q]
        box_val, box_ind = x[..., :n_box].max(-1), x[..., :n_box].argmax(-1)
        allowed = (box_val == aligned) & (box_val > np.float32(-1e5))
        if "rel" in names:
            o = int(starts[names.index("rel")])
            a = allowed.copy()
            a[:, L + 1:] = False                                          # :571
            for b in range(B):
                sel = np.zeros(n_box, dtype=bool)
                sel[box_ind[b][a[b]]] = True
                pair = (sel[:, None] & sel[None, :]).reshape(-1)
                x[b, :, o:o + n_box * n_box][:, ~pair] -= np.float32(100)                      # :580
                x[b, :, o + np.arange(n_box) * (n_box + 1)] = np.float32(-1e10)               # :582
        if "attr" in names:
            o = int(starts[names.index("attr")])
            for b in range(B):
                sel = np.zeros(n_box, dtype=bool)
                sel[box_ind[b][allowed[b]]] = True
                x[b, :, o:o + n_box][:, ~sel] = np.float32(-1e10)                              # :594
    order = np.argsort(-x.astype(np.float64), axis=-1, kind="stable")[..., :5]                # :596
    return dict(logit=x, top5=order, factor2img=factor2img)


def grounding_decode_lists(top5, factor2img, tmask, factor_names, vis_split, vis_box_index=None):
    """The list half of decode_grounding_on_factor (joint.py:598-629): columns -> (factor name, box id | (box, box)),
    rows filtered by the query mask (filter_list, src/utility/fn.py:143-151)."""
    import bisect
    names = [str(n) for n in factor_names]
    widths = [int(w) for w in vis_split]
    starts = [0] + list(np.cumsum(widths))
    out_f, out_i = [], []
    for b in range(len(top5)):
        box_index = list(range(200)) if vis_box_index is None else [int(i) for i in vis_box_index[b]]
        rows_f, rows_i = [], []
        for q in range(len(top5[b])):
            if not tmask[b][q]:
                continue
            cands = []
            for idx in top5[b][q]:
                idx = int(idx)
                grp = bisect.bisect_left(starts, idx)
                if starts[grp] != idx:
                    grp -= 1
                idx -= starts[grp]
                name = names[grp]
                cands.append([name, [box_index[idx // widths[0]], box_index[idx % widths[0]]] if name == "rel" else box_index[idx]])
            rows_f.append(cands)
            rows_i.append(int(factor2img[b][q]))
        out_f.append(rows_f)
        out_i.append(rows_i)
    return out_f, out_i


def _projective_single_root_trees(n_words):
    """All head vectors (heads[c] for c = 1..n_words, 0 = root) that are projective, spanning and
    single-rooted -- the support of both DPs (deptree.py:325-378 states the same predicates)."""
    N = n_words + 1
    for heads in itertools.product(range(N), repeat=n_words):
        h = (-1,) + heads
        if sum(1 for c in range(1, N) if h[c] == 0) != 1:
            continue
        ok = all(h[c] != c for c in range(1, N))
        for c in range(1, N):                      # acyclic / spanning
            seen, x = set(), c
            while ok and x != 0:
                if x in seen:
                    ok = False
                seen.add(x)
                x = h[x]
        if not ok:
            continue
        for c in range(1, N):                      # projective: every word strictly inside an arc descends from its head
            lo, hi = min(c, h[c]), max(c, h[c])
            for m in range(lo + 1, hi):
                x = m
                while x != 0 and x != h[c]:
                    x = h[x]
                if x != h[c]:
                    ok = False
        if ok:
            yield h


def dmv1o_tree_score(dec, attach, heads, length):
    """Score of ONE tree (heads[c] for c = 1..length, 0 = root) under the valence rules of dmv.py:36-62 (see
    enumerate_dmv1o).  Used to compare Viterbi trees by value when ties make the arg-max non-unique."""
    NOCHILD, HASCHILD, LEFT, RIGHT, GO, STOP = 1, 0, 0, 1, 0, 1
    s = 0.0
    for head in range(0, length + 1):
        for direction in (LEFT, RIGHT):
            if head == 0 and direction == LEFT:
                continue
            kids = [c for c in range(1, length + 1) if heads[c] == head and ((c < head) == (direction == LEFT))]
            kids.sort(key=lambda c: -abs(c - head))
            val = NOCHILD
            for c in kids:
                s += float(dec[head, direction, val, GO]) + float(attach[head, c, val])
                val = HASCHILD
            s += float(dec[head, direction, val, STOP])
    return s


def is_projective_tree(heads, length):
    "heads[1..length] form a spanning, single-rooted, projective tree over 0..length."
    return _is_proj([int(x) for x in heads[:length + 1]], length)


def _is_proj(h, n):
    if sum(1 for c in range(1, n + 1) if h[c] == 0) != 1:
        return False
    for c in range(1, n + 1):
        seen, x = set(), c
        while x != 0:
            if x in seen or not (0 <= h[x] <= n) or h[x] == x:
                return False
            seen.add(x)
            x = h[x]
    for c in range(1, n + 1):
        lo, hi = min(c, h[c]), max(c, h[c])
        for m in range(lo + 1, hi):
            x = m
            while x != 0 and x != h[c]:
                x = h[x]
            if x != h[c]:
                return False
    return True


def enumerate_deptree(arc, length):
    """log-sum and max over trees of sum_c arc[head(c), c] (float64)."""
    scores = [sum(float(arc[h[c], c]) for c in range(1, length + 1)) for h in _projective_single_root_trees(length)]
    m = max(scores)
    return m + math.log(sum(math.exp(s - m) for s in scores)), m


def enumerate_dmv1o(dec, attach, length):
    """log-sum and max over trees of the valence-DMV score implied by dmv.py:36-62: for each head
    and direction the children are generated OUTSIDE-IN; the first (outermost) attachment is scored
    with valence NOCHILD, later (inner) ones with HASCHILD, and the STOP decision with the valence
    reached after the last attachment.  The root (index 0) only generates to the right."""
    NOCHILD, HASCHILD, LEFT, RIGHT, GO, STOP = 1, 0, 0, 1, 0, 1
    scores = []
    for h in _projective_single_root_trees(length):
        s = 0.0
        for head in range(0, length + 1):
            for direction in (LEFT, RIGHT):
                if head == 0 and direction == LEFT:
                    continue
                kids = [c for c in range(1, length + 1) if h[c] == head and ((c < head) == (direction == LEFT))]
                kids.sort(key=lambda c: -abs(c - head))           # outermost first
                val = NOCHILD
                for c in kids:
                    s += float(dec[head, direction, val, GO]) + float(attach[head, c, val])
                    val = HASCHILD
                s += float(dec[head, direction, val, STOP])
        scores.append(s)
    m = max(scores)
    return m + math.log(sum(math.exp(s - m) for s in scores)), m


# ----------------------------------------------------------------------------------------------
# Visual encoder's pairwise relation features, src/model/vis_encoder/box_rel.py:29-52 (numpy, fp64): the reference's
# formulation restated literally -- pairwise mean of [box ; mean box] inputs, Linear, LeakyReLU -- and its adjoint.
# ----------------------------------------------------------------------------------------------
def box_rel(feat, weight, bias, slope=0.01, img_feat=True, dout=None):
    """feat [B,R,n], weight [H,n_in], bias [H] -> rel [B,R*R,H]; with dout [B,R*R,H] also (g_feat, g_weight, g_bias)."""
    feat, weight, bias = (np.asarray(x, dtype=np.float64) for x in (feat, weight, bias))
    B, R, n = feat.shape
    inputs = np.concatenate([feat, np.broadcast_to(feat.mean(1, keepdims=True), feat.shape)], -1) if img_feat else feat   # :33-38
    rel_inp = (inputs[:, None, :, :] + inputs[:, :, None, :]) / 2                                                          # :41
    pre = rel_inp @ weight.T + bias                                                                                        # rel_fc.linear
    rel = np.where(pre > 0, pre, pre * slope).reshape(B, R * R, -1)                                                        # LeakyReLU, :45
    if dout is None:
        return rel
    g_pre = np.asarray(dout, dtype=np.float64).reshape(pre.shape) * np.where(pre > 0, 1.0, slope)
    g_bias = g_pre.sum((0, 1, 2))
    g_weight = np.einsum("bijh,bijn->hn", g_pre, rel_inp)
    g_rel_inp = g_pre @ weight
    g_inputs = (g_rel_inp.sum(1) + g_rel_inp.sum(2)) / 2
    if img_feat:
        g_feat = g_inputs[..., :n] + g_inputs[..., n:].sum(1, keepdims=True) / R
    else:
        g_feat = g_inputs
    return rel, g_feat, g_weight, g_bias


# ----------------------------------------------------------------------------------------------
# Encoder projection, src/model/nn/common.py:23-51 (`MLP`: Linear -> LeakyReLU; dropout is the identity at p = 0 / eval) as
# the word / child / parent encoders of src/model/joint.py:270-277 apply it to all token rows (numpy, fp64), and its adjoint.
# ----------------------------------------------------------------------------------------------
def mlp(x, weight, bias=None, slope=0.01, dout=None, branch=None):
    """x [..., n_in], weight [n_out, n_in], bias [n_out] -> y [..., n_out]; slope None = no activation (activate=False).
    With dout [..., n_out] also (g_x, g_weight, g_bias).  branch [..., n_out] bool: take LeakyReLU's positive branch where
    true instead of where pre > 0 -- the adjoint of the function a reduced-precision path evaluated, whose pre-activations
    within rounding of zero can sit on the other side."""
    x, weight = np.asarray(x, dtype=np.float64), np.asarray(weight, dtype=np.float64)
    pre = x @ weight.T                                                              # common.py:48
    if bias is not None:
        pre = pre + np.asarray(bias, dtype=np.float64)
    pos = (pre > 0) if branch is None else np.asarray(branch, dtype=bool)
    y = pre if slope is None else np.where(pos, pre, pre * slope)                   # common.py:49
    if dout is None:
        return y
    g_pre = np.asarray(dout, dtype=np.float64) * (1.0 if slope is None else np.where(pos, 1.0, slope))
    g2, x2 = g_pre.reshape(-1, weight.shape[0]), x.reshape(-1, weight.shape[1])
    return y, g_pre @ weight, g2.T @ x2, g2.sum(0)


def lang_feat(x, lengths, heads, w_word, b_word, w_child, b_child, w_parent, b_parent, w1, w2, b_arc, slope=0.01, dout=None,
              child_branch=None, parent_branch=None):
    """joint.py:262-288 (the feature half of lang_feat_max_tree), fp64: root = masked mean, x = cat([root, x]), word /
    child / parent encoders (parent on x gathered by the predicted heads), arc_repr, txt = cat([word_repr, arc_repr]).
    x [B,L,h], lengths [B], heads [B,L+1] -> txt [B,2(L+1),d]; with dout also the gradients, as a dict by parameter name.
    child_branch / parent_branch [B,L+1,d] bool: LeakyReLU branches to take (see `mlp`)."""
    x = np.asarray(x, dtype=np.float64)
    lengths, heads = np.asarray(lengths), np.asarray(heads)
    B, L, h = x.shape
    N = L + 1
    mask = (np.arange(L)[None] < lengths[:, None])[..., None]
    root = (x * mask).sum(1) / lengths[:, None]                                       # :263-265
    x1 = np.concatenate([root[:, None], x], 1)                                        # :266
    xg = np.take_along_axis(x1, heads[..., None], 1)                                  # :271-273
    word = mlp(x1, w_word, b_word, None)                                              # :267 (activate: false)
    child = mlp(x1, w_child, b_child, slope, branch=child_branch)                     # :269
    parent = mlp(xg, w_parent, b_parent, slope, branch=parent_branch)                 # :270
    arc = arc_encoder(child, parent, w1, w2, b_arc, np.float64)                       # :278-286
    txt = np.concatenate([word, arc], 1)                                              # :288
    if dout is None:
        return txt
    dout = np.asarray(dout, dtype=np.float64)
    d_child, d_parent, d_w1, d_w2, d_b = arc_encoder_backward(child, parent, w1, w2, dout[:, N:], np.float64)
    _, gx_w, g_ww, g_bw = mlp(x1, w_word, b_word, None, dout[:, :N])
    _, gx_c, g_wc, g_bc = mlp(x1, w_child, b_child, slope, d_child, child_branch)
    _, gx_p, g_wp, g_bp = mlp(xg, w_parent, b_parent, slope, d_parent, parent_branch)
    g_x1 = gx_w + gx_c
    for b in range(B):
        np.add.at(g_x1[b], heads[b], gx_p[b])
    g_x = g_x1[:, 1:] + mask * (g_x1[:, :1] / lengths[:, None, None])
    return txt, dict(x=g_x, w_word=g_ww, b_word=g_bw, w_child=g_wc, b_child=g_bc, w_parent=g_wp, b_parent=g_bp, w1=d_w1, w2=d_w2,
                     b_arc=d_b)


def lang_feat_marginal(grad_attach, heads, lengths, add_marginal=True):
    """joint.py:246-262: (txt_marginal [B,2N], txt_mask [B,2N]) from d logZ / d attach [B,N,N,2] and `predicted`."""
    grad_attach, heads, lengths = np.asarray(grad_attach, dtype=np.float64), np.asarray(heads), np.asarray(lengths)
    B, N = heads.shape
    mask = np.concatenate([np.zeros((B, 1), bool), np.arange(N - 1)[None] < lengths[:, None]], 1)   # :248
    arc_margin = grad_attach.sum(-1)                                                  # :255
    am = np.take_along_axis(arc_margin, heads[..., None], -1)[..., 0] if add_marginal else mask.astype(np.float64)   # :258-262
    return np.concatenate([mask.astype(np.float64), am], 1), np.concatenate([mask, mask], 1)


# ----------------------------------------------------------------------------------------------
# Score construction of DiscriminativeNDMV._forward, src/model/ldndmv.py:184-209 (numpy, fp64), from the scorers' projected
# inputs (nn/dmv_spec.py:66-76: einsum('bhdve,bcdve->bhcdv')) to the merged potentials, and its adjoint.
# ----------------------------------------------------------------------------------------------
def ndmv_potentials(x1, x2, y1, y2, root_rule, token, head_mask=None, mask_fill=-1e20, g_mdec=None, g_mattach=None):
    """x1, y1 [B,L,2,2,r]; x2 [T,2,2,r]; y2 [2,2,2,r]; root_rule [T]; token [B,L] -> (merged_dec [B,N,2,2,2], merged_attach
    [B,N,N,2]); with the cotangents also dict(x1, x2, y1, y2, root_rule)."""
    x1, x2, y1, y2, root_rule = (np.asarray(a, dtype=np.float64) for a in (x1, x2, y1, y2, root_rule))
    token = np.asarray(token)
    B, L = token.shape
    T, N = x2.shape[0], L + 1
    score = np.einsum("bhdve,cdve->bhcdv", x1, x2)                                          # dmv_spec.py:73
    smax = score.max(2, keepdims=True)
    lse = smax + np.log(np.exp(score - smax).sum(2, keepdims=True))
    attach_rule = score - lse                                                               # ldndmv.py:185 log_softmax(2)
    bi = np.arange(B)[:, None, None]
    hi = np.arange(L)[None, :, None]
    gathered = attach_rule[bi, hi, token[:, None, :]]                                       # [B,L(h),L(c),2(d),2(v)], :189-190
    left = np.tril(np.ones((L, L)), -1)[None, :, :, None]
    right = np.triu(np.ones((L, L)), 1)[None, :, :, None]
    attach = gathered[..., 0, :] * left + gathered[..., 1, :] * right                       # :191-194
    hm = None if head_mask is None else np.asarray(head_mask, dtype=bool)
    if hm is not None:
        attach = np.where(hm[:, :, None, None], mask_fill, attach)                          # :195-199
    dscore = np.einsum("bhdve,kdve->bhkdv", y1, y2).transpose(0, 1, 3, 4, 2)               # :201 permute(0,1,3,4,2)
    dmax = dscore.max(-1, keepdims=True)
    dec = dscore - (dmax + np.log(np.exp(dscore - dmax).sum(-1, keepdims=True)))
    root = root_rule[token]                                                                 # :205-207
    mdec = np.full((B, N, 2, 2, 2), -1e12)
    matt = np.full((B, N, N, 2), -1e12)                                                     # distributions.py:253-265
    mdec[:, 0, 1] = 0.0
    mdec[:, 1:] = dec
    matt[:, 0, 1:, 1] = root
    matt[:, 1:, 1:] = attach
    if g_mdec is None:
        return mdec, matt
    g_mdec, g_mattach = np.asarray(g_mdec, dtype=np.float64), np.asarray(g_mattach, dtype=np.float64)
    g_attach = g_mattach[:, 1:, 1:].copy()
    if hm is not None:
        g_attach[hm] = 0.0
    g_gath = np.stack([g_attach * left, g_attach * right], -2)                              # [B,L,L,2(d),2(v)]
    g_rule = np.zeros_like(attach_rule)
    for b in range(B):
        for c in range(L):
            g_rule[b, :, token[b, c]] += g_gath[b, :, c]
    g_score = g_rule - np.exp(attach_rule) * g_rule.sum(2, keepdims=True)                   # log_softmax adjoint
    g_dec = g_mdec[:, 1:]
    g_dscore = (g_dec - np.exp(dec) * g_dec.sum(-1, keepdims=True)).transpose(0, 1, 4, 2, 3)  # -> [b,h,k,d,v]
    g_root = np.zeros(T)
    np.add.at(g_root, token, g_mattach[:, 0, 1:, 1])
    return mdec, matt, dict(x1=np.einsum("bhcdv,cdve->bhdve", g_score, x2), x2=np.einsum("bhcdv,bhdve->cdve", g_score, x1),
                            y1=np.einsum("bhkdv,kdve->bhdve", g_dscore, y2), y2=np.einsum("bhkdv,bhdve->kdve", g_dscore, y1),
                            root_rule=g_root)


# ---------------------------------------------------------------- data feed
def feed_kmeans(x, init_centroids, k, max_it=32):
    """ConstantTokenNumSampler.kmeans (datamodule/sampler.py:148-191) from given initial centroids, in the reference's dense
    form: an [n, k] distance matrix per iteration, float32 arithmetic, first-minimum / first-maximum tie rules.
    Returns (centroids of the surviving clusters, cluster id per point renumbered over the survivors)."""
    x = np.asarray(x, np.float32)
    c = np.asarray(init_centroids, np.float32)          # may hold fewer than k entries (fewer distinct values than clusters)
    dist = np.abs(x[:, None] - c[None, :])
    y = dist.argmin(-1)
    d = dist[np.arange(len(x)), y]
    for _ in range(max_it):
        while True:                                      # sampler.py:164-176
            counts = np.bincount(y, minlength=k)
            empty = np.nonzero(counts == 0)[0]
            if len(empty) == 0:
                break
            for e in empty:
                counts = np.bincount(y, minlength=k)
                members = np.nonzero(y == counts.argmax())[0]
                y[members[d[members].argmax()]] = e
        counts = np.bincount(y, minlength=k)
        old = c
        c = (np.bincount(y, weights=x.astype(np.float64), minlength=k).astype(np.float32) / counts.astype(np.float32)).astype(np.float32)
        dist = np.abs(x[:, None] - c[None, :])
        y = dist.argmin(-1)
        d = dist[np.arange(len(x)), y]
        if old.shape == c.shape and (old == c).all():
            break
    alive = np.unique(y)
    return c[alive], np.searchsorted(alive, y)


def feed_batches(seq_len, buckets, chunks, bucket_perms, batch_perm, single_sent_threshold=-1, sort_in_batch=True):
    """One epoch of ConstantTokenNumSampler (sampler.py:86-140): lists of sentence ids."""
    raw = []
    for items, ch, perm in zip(buckets, chunks, bucket_perms):
        at = 0
        for j in range(ch):
            size = (len(items) - j - 1) // ch + 1
            raw.append([items[p] for p in perm[at:at + size]])
            at += size
    out = []
    for r in batch_perm:
        keep = [i for i in raw[r] if single_sent_threshold == -1 or seq_len[i] < single_sent_threshold]
        if sort_in_batch:
            keep.sort(key=lambda i: -seq_len[i])
        if keep:
            out.append(keep)
        out.extend([i] for i in raw[r] if single_sent_threshold != -1 and seq_len[i] >= single_sent_threshold)
    return out
