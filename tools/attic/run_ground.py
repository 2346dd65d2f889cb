"""Grounding loss forward + gradients only (bf16), config-2 and the shipped layout -- for rocprofv3 --kernel-trace --stats."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
torch.autograd.set_multithreading_enabled(False)
import os
for B, V in (((256, 36),) if os.environ.get("VLG_RUN_GROUND_C2") else ((256, 36), (64, 1369))):
    L, d = 40, 128
    Q = 2 * (L + 1)
    g = torch.Generator().manual_seed(0)
    lengths = torch.randint(L // 2, L + 1, (B,), generator=g)
    m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), torch.arange(L)[None] < lengths[:, None]], 1)
    tmask = torch.cat([m1, m1], 1).to(dev)
    vmask = (torch.rand(B, V, generator=g) > 0.1).to(dev)
    marg = (torch.rand(B, Q, generator=g).to(dev) * tmask)
    txt = (torch.randn(B, Q, d, generator=g) * 0.5).to(dev, torch.bfloat16).requires_grad_(True)
    vis = (torch.randn(B, V, d, generator=g) * 0.5).to(dev, torch.bfloat16).requires_grad_(True)
    def ours():
        total, sums = align.grounding_loss_factor_ce(txt, vis, tmask, vmask, marg, int(lengths.sum()), 1.0)
        return torch.autograd.grad(total, [txt, vis])
    for _ in range(3): ours()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ours()
    e1.record(); torch.cuda.synchronize()
    print(f'B={B} V={V}: {e0.elapsed_time(e1) / 20:.3f} ms', flush=True)
