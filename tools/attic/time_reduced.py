"""gather_logit_reduced + caption-image cross-entropy (joint.py:421-432, 493-499), forward + gradients, config-2 shapes:
fused path (no [B,B,Q,V] tensor) vs the same steps as torch ops."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgae_amd import align   # noqa: E402

dev = torch.device("cuda:0")
B, L, V, d = 256, 40, 36, 128
Q = 2 * (L + 1)


def timed(fn, n=10):
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


for dt in (torch.bfloat16, torch.float32):
    torch.manual_seed(0)
    txt = (torch.randn(B, Q, d, device=dev) * 0.5).to(dt).requires_grad_()
    vis = (torch.randn(B, V, d, device=dev) * 0.5).to(dt).requires_grad_()
    m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool, device=dev), torch.ones(B, L, dtype=torch.bool, device=dev)], 1)
    tmask = torch.cat([m1, m1], 1)
    vmask = torch.rand(B, V, device=dev) > 0.1
    marg = torch.rand(B, Q, device=dev) * tmask
    target = torch.arange(B, device=dev)

    def ours():
        logit = align.gather_logit_reduced(None, None, (vis, vmask, None), (txt, tmask, marg), None)
        loss = torch.nn.functional.cross_entropy(logit, target)
        return loss, torch.autograd.grad(loss, [txt, vis])

    def ref():
        att = torch.einsum("bqd,avd->baqv", txt.float(), vis.float())   # fp32 products, like the kernels
        att = att.masked_fill(~vmask[None, :, None, :], -1e20).masked_fill(~tmask[:, None, :, None], -1e20)
        logit = (att.max(-1).values * marg[:, None]).sum(-1) / marg.sum(1, keepdim=True)
        loss = torch.nn.functional.cross_entropy(logit, target)
        return loss, torch.autograd.grad(loss, [txt, vis])

    (l1, g1), (l2, g2) = ours(), ref()
    print(dt, "loss", float(l1), float(l2), "max grad diff", max(float((a.float() - b.float()).abs().max()) for a, b in zip(g1, g2)),
          "grad scale", float(g2[0].float().abs().max()))
    print(f"  ours fwd+bwd {timed(ours):.3f} ms   torch ops fwd+bwd {timed(ref):.3f} ms")
