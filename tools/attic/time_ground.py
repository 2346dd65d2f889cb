"""Grounding loss forward + gradients at config-2 (B = A = 256, Q = 82, V = 36, d = 128; `--shipped`: B = 64, V = 1369):
vlg_grounding_loss vs the
reference's formulation in torch ops (einsum -> masked_fill -> max -> log_softmax -> diagonal; autograd).
Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, L, V, d = 256, 40, 36, 128
if len(sys.argv) > 1 and sys.argv[1] == '--shipped':   # the shipped factor layout: obj 36 + rel 1296 + attr 36 + img 1, batch 64
    B, V = 64, 1369
Q = 2 * (L + 1)
g = torch.Generator().manual_seed(0)
lengths = torch.randint(L // 2, L + 1, (B,), generator=g)
m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), torch.arange(L)[None] < lengths[:, None]], 1)
tmask = torch.cat([m1, m1], 1).to(dev)
vmask = (torch.rand(B, V, generator=g) > 0.1).to(dev)
marg = (torch.rand(B, Q, generator=g).to(dev) * tmask)
num = int(lengths.sum())
for dt in (torch.bfloat16, torch.float32):
    txt = (torch.randn(B, Q, d, generator=g) * 0.5).to(dev, dt).requires_grad_(True)
    vis = (torch.randn(B, V, d, generator=g) * 0.5).to(dev, dt).requires_grad_(True)
    def ours():
        total, sums = align.grounding_loss_factor_ce(txt, vis, tmask, vmask, marg, num, 1.0)
        return total, torch.autograd.grad(total, [txt, vis])
    def ref():
        att = torch.einsum("avd,bqd->baqv", vis.float(), txt.float())
        att = att.masked_fill(~vmask[None, :, None, :], -1e20).masked_fill(~tmask[:, None, :, None], -1e20)
        lv = att.max(3).values.log_softmax(1)
        t2v = -(lv.diagonal().T * marg).sum()
        lq = att.max(2).values.log_softmax(0)
        v2t = -(lq.diagonal().T * vmask).sum()
        total = t2v / (t2v.detach() + 1e-6) * num + v2t / (v2t.detach() + 1e-6) * num
        return total, torch.autograd.grad(total, [txt, vis])
    (t1, g1), (t2, g2) = ours(), ref()
    print(dt, 'total', float(t1), float(t2), 'max grad diff', max(float((a.float() - b.float()).abs().max()) for a, b in zip(g1, g2)),
          'grad scale', float(g2[0].abs().max()))
    for name, fn, n in (('vlg_grounding_loss fwd+bwd', ours, 20), ('torch ops fwd+bwd', ref, 5)):
        for _ in range(2): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        print(f'  {name}: {e0.elapsed_time(e1) / n:.3f} ms')
