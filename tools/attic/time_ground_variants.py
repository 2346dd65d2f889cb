"""Kernel-time A/B of grounding-loss library variants in tools/_v/gb_*.so (run each child under rocprofv3)."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, L, V, d = 256, 40, 36, 128
Q = 2 * (L + 1)
g = torch.Generator().manual_seed(0)
lengths = torch.randint(L // 2, L + 1, (B,), generator=g)
m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), torch.arange(L)[None] < lengths[:, None]], 1)
tmask = torch.cat([m1, m1], 1).to(dev)
vmask = (torch.rand(B, V, generator=g) > 0.1).to(dev)
marg = (torch.rand(B, Q, generator=g).to(dev) * tmask)
dt = torch.bfloat16 if len(sys.argv) > 1 and sys.argv[1] == 'bf16' else torch.float32
txt = (torch.randn(B, Q, d, generator=g) * 0.5).to(dev, dt).requires_grad_(True)
vis = (torch.randn(B, V, d, generator=g) * 0.5).to(dev, dt).requires_grad_(True)
for _ in range(5):
    total, sums = align.grounding_loss_factor_ce(txt, vis, tmask, vmask, marg, int(lengths.sum()), 1.0)
    torch.autograd.grad(total, [txt, vis])
torch.cuda.synchronize()
