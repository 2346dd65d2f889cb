"""Fused score construction (vlgae_amd.scorer) vs the reference's formulation in torch ops, B=256 L=40 T=45 r=16: forward and forward + DP + backward."""
import sys, time, torch
sys.path.insert(0, '.')
import vlgae_amd.torch_struct as ts
from vlgae_amd import scorer
dev = torch.device('cuda:0')
B, L, T, r = 256, 40, 45, 16
g = torch.Generator().manual_seed(9)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
ins = [rnd(B, L, 2, 2, r, sc=0.5).requires_grad_(True), rnd(T, 2, 2, r, sc=0.5).requires_grad_(True), rnd(B, L, 2, 2, r, sc=0.5).requires_grad_(True),
       rnd(2, 2, 2, r, sc=0.5).requires_grad_(True), torch.randn(T, generator=g).log_softmax(-1).to(dev).requires_grad_(True)]
token = torch.randint(0, T, (B, L), generator=g).to(dev)
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
def ours():
    md, ma = scorer.ndmv_potentials(*ins, token)
    return torch.autograd.grad(ts.DMV1o([md, ma], lengths).partition.sum(), ins)
def ref_ops():   # ldndmv.py:185-209 with torch ops, DMV1o ours
    x1, x2, y1, y2, root_rule = ins
    attach_rule = torch.einsum("bhdve,cdve->bhcdv", x1, x2).log_softmax(2)
    attach_prob = attach_rule.gather(2, token.reshape(B, 1, L, 1, 1).expand(B, L, L, 2, 2))
    left = torch.tril(torch.ones(L, L, device=dev), diagonal=-1); right = torch.triu(torch.ones(L, L, device=dev), diagonal=1)
    attach_prob = attach_prob[..., 0, :] * left.unsqueeze(0).unsqueeze(-1) + attach_prob[..., 1, :] * right.unsqueeze(0).unsqueeze(-1)
    dec = torch.einsum("bhdve,kdve->bhkdv", y1, y2).permute(0, 1, 3, 4, 2).log_softmax(-1)
    root = torch.gather(root_rule.unsqueeze(0).expand(B, -1), 1, token)
    md, ma = ts.DMV1o.merge(dec, attach_prob, root)
    return torch.autograd.grad(ts.DMV1o([md, ma], lengths).partition.sum(), ins)
def wall(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
a, b = ours(), ref_ops()
print('max grad diff', [float((x - y).abs().max()) for x, y in zip(a, b)])
with torch.autograd.set_multithreading_enabled(False):
    print('fused scorer + DP fwd+bwd: %.3f ms; torch-op glue + DP: %.3f ms' % (wall(ours), wall(ref_ops)))
