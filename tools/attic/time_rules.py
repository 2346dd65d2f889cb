"""Rule-table DP (vlg_dmv1o_rules) vs the same work done the reference's way on the GPU
(gather + masks + merge in PyTorch ops, then the merged-potential kernel)."""
import sys, torch
sys.path.insert(0, '.')
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
B, L, T = 256, 40, 45
g = torch.Generator().manual_seed(0)
ar = torch.randn(B, L, T, 2, 2, generator=g).log_softmax(2).to(dev).requires_grad_()
dc = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev).requires_grad_()
rr = torch.randn(T, generator=g).log_softmax(-1).to(dev).requires_grad_()
tok = torch.randint(0, T, (B, L), generator=g).to(dev)
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
left = torch.tril(torch.ones(L, L, device=dev), -1); right = torch.triu(torch.ones(L, L, device=dev), 1)
def fused():
    z = ts.DMV1oRules(ar, dc, rr, tok, lengths).partition.sum()
    return torch.autograd.grad(z, [ar, dc, rr])
def unfused():   # ldndmv.py:189-209 with torch ops, then the merged kernel
    ap = ar.gather(2, tok.reshape(B, 1, L, 1, 1).expand(B, L, L, 2, 2))
    ap = ap[..., 0, :] * left[None, :, :, None] + ap[..., 1, :] * right[None, :, :, None]
    root = torch.gather(rr.expand(B, -1), 1, tok)
    md, ma = ts.DMV1o.merge(dc, ap, root)
    z = ts.DMV1o([md, ma], lengths).partition.sum()
    return torch.autograd.grad(z, [ar, dc, rr])
a, b = fused(), unfused()
print('max |grad diff|', [float((x - y).abs().max()) for x, y in zip(a, b)])
for name, fn in (('fused rules kernel', fused), ('torch glue + merged kernel', unfused)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{name}: {e0.elapsed_time(e1) / 30 * 1e3:.0f} us per forward+backward')
