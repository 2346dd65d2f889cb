"""Grounding decoder (joint.py:512-596, tensor half): fused alignment outputs + vlg_grounding_decode vs the same steps as
torch ops on the materialised [B,A,Q,V] tensor.  Two layouts: config-2 (B=256, one factor of 36 boxes) and the shipped
factor layout obj 36 + rel 1296 + attr 36 + img 1 at B=64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgae_amd import align   # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def torch_steps(txt, vis, tmask, vmask, pen, seg, n_box, rel_off, attr_off, L):
    att = torch.einsum("bqd,avd->baqv", txt.float(), vis.float())
    att = att.masked_fill(~vmask[None, :, None, :], -1e20).masked_fill(~tmask[:, None, :, None], -1e20)
    f2i = att.max(3).values.max(1).indices
    x = att.diagonal(dim1=0, dim2=1).permute(2, 0, 1).clone()
    x -= pen.gather(2, seg.long()[None, None, :].expand(x.shape[0], x.shape[1], -1))
    aligned = x.max(-1).values
    bval, bind = x[..., :n_box].max(-1)
    allowed = (bval == aligned) & (bval > -1e5)
    B, Q, _ = x.shape
    for off, rows, pair in ((rel_off, L + 1, True), (attr_off, Q, False)):
        if off < 0:
            continue
        a = allowed.clone()
        a[:, rows:] = False
        sel = torch.zeros(B, n_box, dtype=torch.bool, device=x.device)
        sel[torch.arange(B, device=x.device)[:, None].expand(B, Q)[a], bind[a]] = True
        if pair:
            ok = (sel[:, :, None] & sel[:, None, :]).view(B, 1, -1)
            blk = x[..., off:off + n_box * n_box]
            blk -= 100.0 * (~ok)
            blk.view(B, Q, n_box, n_box).diagonal(dim1=2, dim2=3).fill_(-1e10)
        else:
            x[..., off:off + n_box].masked_fill_(~sel[:, None, :], -1e10)
    return x.argsort(-1, descending=True)[..., :5], f2i


for B, L, split, names in ((256, 40, [36], ["obj"]), (64, 40, [36, 1296, 36, 1], ["obj", "rel", "attr", "img"])):
    torch.manual_seed(0)
    V, Q, d = sum(split), 2 * (L + 1), 128
    txt = (torch.randn(B, Q, d, device=dev) * 0.5).bfloat16()
    vis = (torch.randn(B, V, d, device=dev) * 0.5).bfloat16()
    m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool, device=dev), torch.ones(B, L, dtype=torch.bool, device=dev)], 1)
    tmask = torch.cat([m1, m1], 1)
    vmask = torch.rand(B, V, device=dev) > 0.1
    tag = torch.randint(0, 6, (B, L), device=dev)
    pos_for = dict(obj=torch.tensor([0, 1]), rel=torch.tensor([1, 2]), attr=torch.tensor([5]))
    pen, seg = align.grounding_prior(tag, names, split, pos_for, Q, scale=1e10)
    start = [0]
    for w in split:
        start.append(start[-1] + w)
    rel = start[names.index("rel")] if "rel" in names else -1
    attr = start[names.index("attr")] if "attr" in names else -1
    ours = lambda: align.grounding_decode(txt, vis, tmask, vmask, pen, seg, True, split[0], rel, attr, L + 1)
    ref = lambda: torch_steps(txt, vis, tmask, vmask, pen, seg, split[0], rel, attr, L)
    r, (top_t, f2i_t) = ours(), ref()
    same = (r["top5"].long() == top_t).float().mean().item()
    print(f"B={B} Q={Q} V={V}: ours {timed(ours) * 1e3:.0f} us   torch ops {timed(ref) * 1e3:.0f} us   "
          f"top-5 agreement {same:.4f}  image agreement {(r['factor2img'].long() == f2i_t).float().mean().item():.4f}")
