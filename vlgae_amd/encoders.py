"""The two trainable encoders between the frozen features and the structured step -- host-side mirror of what
`JointModelBase.forward` runs first (src/model/base.py:229,235; paths relative to /root/reference):

    mlp_encoder(...)           <->  MLPEncoder.forward, src/model/text_encoder/mlp_encoder.py:36-40
                                    x = linear(shared_dropout(dropout(emb)))   (Linear 800 -> 256 without bias, nn.Dropout p = 0.33,
                                    `shared_dropout: 0` in config/model/vlgae.yaml:21-25)
    vis_box_rel_encoder(...)   <->  VisBoxRelSimpleEncoder.forward, src/model/vis_encoder/box_rel.py:29-52 (img_feat: inputs =
                                    [box ; mean_r box]; box_fc / rel_fc / attr_fc = MLP: Linear -> LeakyReLU, nn/common.py:23-51)
                                    followed by the concatenation `vis_feat_unprune` makes of its outputs (src/model/joint.py:137-171):
                                    ONE [B, V, H] tensor with the factors' rows where that method puts them (obj | rel | attr | img)

Both are plain Linear layers: the GEMMs over the ~10^4 token / region rows are library calls (hipBLASLt through torch.mm), the weight
gradients run on the split-K kernel (vlg_linear_wgrad), and everything around them is one HIP pass per stage (csrc/vlg_encoders.hip):
  * the embedding dropout is a counter-based draw (Philox4x32-10 keyed by a device-resident (seed, step) pair, `DeviceRng`): no mask
    tensor is written or read, the adjoint regenerates the bits, a captured HIP graph draws fresh masks on every replay; tests pass the
    reference's recorded masks explicitly instead;
  * the visual encoder never builds the [B,R,R,4096] pairwise-mean tensor (5.1 GB at B = 256) nor the R-fold repeated image half of its
    input: W [x_r ; m] = W_a x_r + W_b m, so P = X W_a^T is one GEMM for all F encoders, C = mean_r(X) W_b^T + b one row per image,
    rel[b,i,j] = LeakyReLU((P_i + P_j) / 2 + C) (657 GFLOP as the reference writes it at B = 256 -> 29 GFLOP);
  * the adjoint of the per-image term is the segment sum of dP, so the [F H, 2n] weight gradient is two split-K products written into
    the two column halves of ONE tensor in place.
No CPU path: tensors must live on the GPU (see _C.require_gpu).
"""
import torch
from torch.autograd.function import once_differentiable

from . import _C
from .align import WgradGroup, linear, linear_wgrad

SLOPE = 0.01   # nn.LeakyReLU() default (nn/common.py:31)
SITE_TEXT_ENCODER, SITE_MID_FF, SITE_SHARED, SITE_SHARED_FF = 1, 2, 3, 4   # which dropout layer of a step draws from the shared DeviceRng state


class DeviceRng:
    """A (seed, step) pair in device memory: the state of the counter-based dropout draws.  `advance()` is a one-thread launch
    (captured into a HIP graph like any other): every replay of a captured step sees a new `step` and therefore new masks, while the
    forward and backward passes of ONE step regenerate identical bits from the same pair."""

    def __init__(self, seed, device):
        self.state = torch.tensor([int(seed), 0], dtype=torch.int64, device=device)

    def advance(self):
        _C.check(_C.lib().vlg_rng_advance(_C.ptr(self.state), _C.stream_of(self.state)), "rng_advance")


def dropout_mask(rng, site, p, n=None, out=None):
    """Explicit float32 masks (0 or 1/(1-p)) from the counter-based generator, n values or into `out` (float32, contiguous): the small
    SharedDropout layers of a step (rows of [B, d]) in one launch (vlg_dropout_mask)."""
    if out is None:
        out = torch.empty((int(n),), dtype=torch.float32, device=rng.state.device)
    _C.check(_C.lib().vlg_dropout_mask(_C.ptr(rng.state), int(site), float(p), _C.ptr(out), out.numel(), _C.stream_of(out)), "dropout_mask")
    return out


def _adt(t):
    if t.dtype == torch.bfloat16:
        return _C.BF16
    if t.dtype == torch.float32:
        return _C.F32
    raise ValueError(f"vlgae_amd.encoders: float32 or bfloat16 tensors, got {t.dtype}")


def _dropout_launch(x2, mask, shared_rows, rng, site, p, add, out):
    rows, cols = x2.shape
    _C.check(_C.lib().vlg_dropout(_C.ptr(x2), _C.ptr(mask), int(shared_rows), None if rng is None else _C.ptr(rng.state), int(site), float(p),
                                  _C.ptr(add), _C.ptr(out), rows, cols, _adt(x2), _adt(out), _C.stream_of(x2)), "dropout")
    return out


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, shared_rows, rng, site, p):
        x2 = x.detach().reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        out = torch.empty_like(x2)
        _dropout_launch(x2, mask, shared_rows, rng, site, p, None, out)
        ctx.args = (mask, shared_rows, rng, site, p)
        return out.view(x.shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mask, shared_rows, rng, site, p = ctx.args
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        return _dropout_launch(g2, mask, shared_rows, rng, site, p, None, torch.empty_like(g2)).view(g.shape), None, None, None, None, None


def dropout(x, p, mask=None, rng=None, site=0, shared_rows=0):
    """nn.Dropout / SharedDropout as one pass: x [..., cols] * m, m = 0 or 1/(1-p).
    mask: explicit float32 values, x's shape flattened to [rows, cols] -- or [rows / shared_rows, cols] with shared_rows > 0 (one mask row
    per `shared_rows` consecutive rows: SharedDropout's [B,1,cols], nn/dropout.py:42-63); rng: a DeviceRng (counter-based draw; `site`
    separates the dropout layers that share one state).  Exactly one of the two."""
    _C.require_gpu(x, "dropout")
    if (mask is None) == (rng is None):
        raise ValueError("dropout: exactly one of mask / rng")
    if mask is not None:
        mask = mask.detach().to(torch.float32).reshape(-1, x.shape[-1]).contiguous()
        rows = x.numel() // x.shape[-1]
        if mask.shape[0] * (shared_rows or 1) != rows:
            raise ValueError(f"dropout: mask rows {mask.shape[0]} x shared_rows {shared_rows or 1} != {rows} rows of x")
    return _Dropout.apply(x, mask, shared_rows, rng, site, p)


class _DropoutLinear(torch.autograd.Function):
    """Linear_nobias(Dropout_p(x)) with the counter-based draw: forward = the dropout pass + the library product; the adjoint's input gradient
    (g @ weight) * keep is ONE launch (align.linear_kn with the draw in its epilogue) instead of a product, a mask pass and their [rows, E] round trip."""

    @staticmethod
    def forward(ctx, x, weight, rng, site, p):
        x2 = x.detach().reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        xd = _dropout_launch(x2, None, 0, rng, site, p, None, torch.empty_like(x2))
        ctx.save_for_backward(xd, weight.detach())
        ctx.args = (rng, site, p, x.shape)
        return torch.nn.functional.linear(xd, weight.detach()).view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        from .align import _wgrad_ok, linear_kn
        xd, weight = ctx.saved_tensors
        rng, site, p, shape = ctx.args
        g2 = g.reshape(-1, g.shape[-1])
        if not g2.is_contiguous():
            g2 = g2.contiguous()
        dx = linear_kn(g2, weight, rng=rng, site=site, p=p).view(shape) if ctx.needs_input_grad[0] else None
        dw = None
        if ctx.needs_input_grad[1]:
            if _wgrad_ok(xd.shape[0], g2.shape[1], xd.shape[1], g2.dtype):
                dw, _ = linear_wgrad(g2, xd, want_bias=False, out_dtype=weight.dtype)
            else:
                dw = g2.t() @ xd
        return dx, dw, None, None, None


def mlp_encoder(emb, weight, p=0.33, mask=None, rng=None, shared_mask=None, training=True):
    """`MLPEncoder.forward` (mlp_encoder.py:36-40): emb [B,L,E] -> x [B,L,n_hidden] = Linear_nobias(SharedDropout(Dropout_p(emb))).
    weight: `linear.weight` [n_hidden, E].  Training mode draws nn.Dropout's mask from `rng` (DeviceRng) or takes it as `mask`
    [B,L,E] (0 or 1/(1-p)); shared_mask [B,E] is the SharedDropout one (`shared_dropout` > 0 in the encoder's config; the shipped
    vlgae.yaml has 0 = Identity).  Eval mode (training=False) or p == 0: the Linear alone."""
    _C.require_gpu(emb, "mlp_encoder")
    x = emb
    if training and p > 0 and rng is not None and mask is None and shared_mask is None and emb.dtype == weight.dtype:
        from . import align
        rows = emb.numel() // emb.shape[-1]   # (the conditions of align.linear_kn_ok on the adjoint's cotangent [rows, 256] and this weight)
        if (not align._KN_LIBRARY and weight.dtype == torch.bfloat16 and weight.dim() == 2 and weight.shape[0] == 256 and rows >= 2048
                and weight.shape[1] % 8 == 0 and weight.stride(1) == 1):
            return _DropoutLinear.apply(emb, weight, rng, SITE_TEXT_ENCODER, p)
    if training and p > 0:
        x = dropout(x, p, mask=mask, rng=rng, site=SITE_TEXT_ENCODER)
    if training and shared_mask is not None:
        x = dropout(x, 0.0, mask=shared_mask, shared_rows=emb.shape[1])
    return linear(x, weight)


# ----------------------------------------------------------------------------------------------------------------------------
def factor_layout(R, add_rel=True, add_attr=True, add_image=True):
    """Row offsets of the factors inside the [B, V, H] tensor `vis_feat_unprune` concatenates (joint.py:143-170: obj, rel, attr, img in
    that order) -> (offsets dict with -1 for absent factors, V, split list, factor names)."""
    off, o, split, names = dict(box=0, rel=-1, attr=-1, img=-1), R, [R], ["obj"]
    if add_rel:
        off["rel"], o = o, o + R * R
        split.append(R * R)
        names.append("rel")
    if add_attr:
        off["attr"], o = o, o + R
        split.append(R)
        names.append("attr")
    if add_image:
        off["img"], o = o, o + 1
        split.append(1)
        names.append("img")
    return off, o, split, names


def factor_mask(box_mask, add_rel=True, add_attr=True, add_image=True):
    """`vis_feat_unprune`'s mask (joint.py:140-170) for the layout above: box_mask [B,R] bool -> [B,V] bool.  The relation mask is the
    strict upper triangle of the outer product of the box mask (:148-152; the branch the shipped data takes: `vis_rel_mask` is not None)."""
    B, R = box_mask.shape
    parts = [box_mask]
    if add_rel:
        parts.append((box_mask.unsqueeze(1) & box_mask.unsqueeze(2)).triu(1).reshape(B, R * R))
    if add_attr:
        parts.append(box_mask)
    if add_image:
        parts.append(torch.ones(B, 1, dtype=torch.bool, device=box_mask.device))
    return torch.cat(parts, 1)


class _VisEncoder(torch.autograd.Function):
    """(box_feat [B,R,n], W [F H, 2n], b [F H]) -> mid [B, V, H]; F = 1 + add_rel + add_attr encoders stacked box | rel | attr."""

    @staticmethod
    def forward(ctx, feat, W, bias, layout, slope):
        off, V = layout
        B, R, n = feat.shape
        FH = W.shape[0]
        F = 1 + (off["rel"] >= 0) + (off["attr"] >= 0)
        H = FH // F
        act = feat.dtype
        x2 = feat.detach().reshape(B * R, n)
        Wd = W.detach().to(act)
        Wa, Wb = Wd[:, :n], Wd[:, n:]
        xm = torch.empty((B, n), dtype=act, device=feat.device)                  # mean over the image's boxes (box_rel.py:37: all R rows)
        _C.check(_C.lib().vlg_ff_context_mean(_C.ptr(x2), _adt(x2), B, R, n, _C.ptr(xm), _adt(xm), _C.stream_of(x2)), "context_mean")
        P = torch.mm(x2, Wa.t())                                                 # [B R, F H]: ONE GEMM for the F encoders
        C = torch.addmm(bias.detach().to(act), xm, Wb.t())                       # [B, F H]: the image half + bias, once per image
        mid = torch.empty((B, V, H), dtype=act, device=feat.device)
        cols = dict(box=0, rel=H if off["rel"] >= 0 else -1, attr=(F - 1) * H if off["attr"] >= 0 else -1)
        _C.check(_C.lib().vlg_vis_encoder(_C.ptr(P), _C.ptr(C), B, R, H, V, FH, cols["box"], cols["rel"], cols["attr"], off["box"], off["rel"],
                                          off["attr"], off["img"], _adt(P), float(slope), _C.ptr(mid), _C.stream_of(P)), "vis_encoder")
        ctx.save_for_backward(x2, xm, P, C, Wd)
        ctx.meta = (off, V, cols, B, R, n, H, FH, float(slope), W.dtype, bias.dtype, feat.dtype)
        return mid

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x2, xm, P, C, Wd = ctx.saved_tensors
        off, V, cols, B, R, n, H, FH, slope, w_dt, b_dt, f_dt = ctx.meta
        act = P.dtype
        g = g.to(act)
        if not g.is_contiguous():
            g = g.contiguous()
        dP, dC = torch.empty_like(P), torch.empty_like(C)
        _C.check(_C.lib().vlg_vis_encoder_backward(_C.ptr(P), _C.ptr(C), _C.ptr(g), B, R, H, V, FH, cols["box"], cols["rel"], cols["attr"], off["box"],
                                                   off["rel"], off["attr"], off["img"], _adt(P), slope, _C.ptr(dP), _C.ptr(dC), _C.stream_of(P)),
                 "vis_encoder_backward")
        dW = db = dx = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if act in (torch.bfloat16, torch.float32):   # both halves of the [F H, 2n] gradient written in place by the split-K kernel
                out_dt = w_dt if w_dt in (torch.float32, torch.bfloat16) else torch.float32
                dW = torch.empty((FH, 2 * n), dtype=out_dt, device=P.device)
                db = torch.empty((FH,), dtype=out_dt, device=P.device)
                wg = WgradGroup(lazy=True)                                   # the two products as one grid per kernel image, the two reductions as one launch
                linear_wgrad(dP, x2, want_bias=False, out=(dW[:, :n], None), defer=wg)
                linear_wgrad(dC, xm, want_bias=True, out=(dW[:, n:], db), defer=wg)
                wg.flush()
                db = db.to(b_dt)
            else:                       # (other types: the library)
                dW = torch.cat([dP.t() @ x2, dC.t() @ xm], 1).to(w_dt)
                db = dC.sum(0).to(b_dt)
        if ctx.needs_input_grad[0]:     # the region features are frozen inputs in training (no gradient asked); parity tests ask
            dx = (dP @ Wd[:, :n]).view(B, R, n) + ((dC @ Wd[:, n:]) / R).unsqueeze(1)
            dx = dx.to(f_dt)
        return dx, dW, db, None, None


def vis_box_rel_encoder(vis_box_feat, weight, bias, add_rel=True, add_attr=True, add_image=True, negative_slope=SLOPE):
    """`VisBoxRelSimpleEncoder.forward` (box_rel.py:29-52; img_feat = True, dropout = 0, use_img = False: the shipped
    config/model/vlgae.yaml:27-35) + the concatenation of `vis_feat_unprune` (joint.py:143-171).
    vis_box_feat [B,R,n]; weight [F H, 2n] / bias [F H]: the `linear` parameters of box_fc, then rel_fc (add_rel), then attr_fc (add_attr)
    stacked along the output dimension (each [H, 2n]: columns [0,n) act on the box, [n,2n) on the image mean).
    Returns (mid [B,V,H], split, factor names): mid's rows are obj | rel (row i R + j = pair (i,j)) | attr | img (= mean_r of the obj rows,
    joint.py:163), i.e. `_mid` of `vis_feat_unprune(..., return_mid=True)`; `factor_mask` builds the matching [B,V] mask."""
    _C.require_gpu(vis_box_feat, "vis_box_rel_encoder")
    B, R, n = vis_box_feat.shape
    F = 1 + bool(add_rel) + bool(add_attr)
    if weight.shape[1] != 2 * n or weight.shape[0] % F or bias.shape[0] != weight.shape[0]:
        raise ValueError(f"vis_box_rel_encoder: weight {tuple(weight.shape)} / bias {tuple(bias.shape)} do not stack {F} encoders [H, {2 * n}]")
    off, V, split, names = factor_layout(R, add_rel, add_attr, add_image)
    feat = vis_box_feat if vis_box_feat.is_contiguous() else vis_box_feat.contiguous()
    return _VisEncoder.apply(feat, weight, bias, (off, V), float(negative_slope)), split, names


def stack_vis_encoder_params(box_fc, rel_fc=None, attr_fc=None):
    """(weight, bias) pairs of the encoder's `MLP.linear` layers -> the stacked tensors `vis_box_rel_encoder` takes."""
    ps = [p for p in (box_fc, rel_fc, attr_fc) if p is not None]
    return torch.cat([w for w, _ in ps], 0), torch.cat([b for _, b in ps], 0)
