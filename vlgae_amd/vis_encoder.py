"""Visual encoder pieces on the hot path's input side -- host-side mirror of the reference interface.

    rel_features(...)  <->  the `rel` output of VisBoxRelSimpleEncoder.forward (src/model/vis_encoder/box_rel.py:29-52)

The reference materialises the pairwise mean of the region inputs, [B,R,R,n_in] with n_in = 4096 (5.1 GB at B = 256), and
runs rel_fc's Linear over it (657 GFLOP at B = 256).  The Linear is linear, so
    Linear((x_i + x_j) / 2) = (W x_i + W x_j) / 2 + b:
one [B R, n_in] x [n_in, H] library GEMM (hipBLASLt through torch.matmul, like every plain projection of the reference) and a
HIP kernel for the broadcast-add + LeakyReLU epilogue and its adjoint (vlg_box_rel_pairwise*, vlgae_amd/csrc/vlg_rel.hip).
"""
import torch
from torch.autograd.function import once_differentiable

from . import _C


class _PairwiseRel(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, bias, slope):
        dt, y_c = _C.in_dtype(y)
        B, R, H = y_c.shape
        b_c = None if bias is None else bias.detach().to(torch.float32).contiguous()
        out = torch.empty((B, R, R, H), dtype=y_c.dtype, device=y_c.device)
        _C.check(_C.lib().vlg_box_rel_pairwise(_C.ptr(y_c), _C.ptr(b_c), B, R, H, dt, float(slope), _C.ptr(out), _C.stream_of(y_c)),
                 "box_rel_pairwise")
        ctx.save_for_backward(y_c, b_c)
        ctx.meta = (dt, float(slope), y.dtype, None if bias is None else bias.dtype)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        y_c, b_c = ctx.saved_tensors
        dt, slope, y_dtype, b_dtype = ctx.meta
        B, R, H = y_c.shape
        g = g.to(y_c.dtype).contiguous()
        want_b = b_c is not None and ctx.needs_input_grad[1]
        nbytes = _C.lib().vlg_box_rel_pairwise_backward_workspace(B, R, H) if want_b else 0
        (g_y, g_b), ws = _C.alloc_f32(y_c.device, ((B, R, H), (H,) if want_b else None), nbytes)
        _C.check(_C.lib().vlg_box_rel_pairwise_backward(_C.ptr(y_c), _C.ptr(b_c), _C.ptr(g), B, R, H, dt, slope, _C.ptr(ws), nbytes,
                                                        _C.ptr(g_y), _C.ptr(g_b), _C.stream_of(y_c)), "box_rel_pairwise_backward")
        return g_y.to(y_dtype), (g_b.to(b_dtype) if want_b else None), None


def pairwise_rel(y, bias=None, negative_slope=0.01):
    """rel[b,i,j,:] = LeakyReLU((y[b,i] + y[b,j]) / 2 + bias)  for y [B,R,H]; returns [B,R,R,H] in y's dtype (fp32 / bf16)."""
    _C.require_gpu(y, "pairwise_rel")
    return _PairwiseRel.apply(y, bias, float(negative_slope))


def rel_features(vis_box_feat, rel_fc_weight, rel_fc_bias, img_feat=True, negative_slope=0.01):
    """The `rel` entry of VisBoxRelSimpleEncoder.forward (box_rel.py:31-45): vis_box_feat [B,R,n] -> [B, R*R, H].
    rel_fc_weight [H, n_in], rel_fc_bias [H] are `rel_fc.linear`'s parameters (n_in = 2n with img_feat, the shipped setting)."""
    feat = vis_box_feat
    B, R, n = feat.shape
    if img_feat:   # box_rel.py:33-38: inputs = [box ; mean over the image's boxes]
        w_box, w_img = rel_fc_weight[:, :n], rel_fc_weight[:, n:]
        # W [x ; m] = W_box x + W_img m: the image half is one row per image instead of R identical ones
        y = torch.matmul(feat, w_box.t()) + torch.matmul(feat.mean(1, keepdim=True), w_img.t())
    else:
        y = torch.matmul(feat, rel_fc_weight.t())
    return pairwise_rel(y, rel_fc_bias, negative_slope).view(B, R * R, -1)
