"""Grounding loss forward + gradients, bf16, config-2 (event-timed; per-kernel times: tools/prof_any.sh)."""
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
vmask = torch.ones(B, V, dtype=torch.bool, device=dev)
marg = (torch.rand(B, Q, generator=g).to(dev) * tmask)
num = int(lengths.sum())
txt = (torch.randn(B, Q, d, generator=g) * 0.5).to(dev, torch.bfloat16).requires_grad_(True)
vis = (torch.randn(B, V, d, generator=g) * 0.5).to(dev, torch.bfloat16).requires_grad_(True)
def ours():
    total, sums = align.grounding_loss_factor_ce(txt, vis, tmask, vmask, marg, num, 1.0)
    return torch.autograd.grad(total, [txt, vis])
for _ in range(5): ours()
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): ours()
e1.record(); torch.cuda.synchronize()
print('grounding loss forward + both gradients: %.3f ms' % (e0.elapsed_time(e1) / 30))
ga, gb = ours()
torch.cuda.synchronize()
import hashlib
print('sha256(d_txt|d_vis) %s   sum|d_txt| %.6f sum|d_vis| %.6f' % (hashlib.sha256(ga.float().cpu().numpy().tobytes() + gb.float().cpu().numpy().tobytes()).hexdigest()[:16], float(ga.float().abs().sum()), float(gb.float().abs().sum())))
