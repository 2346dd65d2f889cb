"""Host-side cost of the DP's API path: cProfile of DMV1o(...).partition.sum() + autograd.grad at B=256 L=40."""
import cProfile, pstats, sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgae_amd.torch_struct as ts
from vlgae_amd.torch_struct import functional as Fn
dev = torch.device("cuda:0")
B, L = 256, 40
g = torch.Generator().manual_seed(1)
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
attach = torch.randn(B, L, L, 2, generator=g).to(dev)
root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
md, ma = md.bfloat16().contiguous(), ma.bfloat16().contiguous()
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
d_, a_ = md.clone().requires_grad_(True), ma.clone().requires_grad_(True)
def api():
    return torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
def raw():
    return Fn.dmv1o_run(md, ma, lengths, 0, True)
for name, fn in (("raw", raw), ("api", api)):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s: host enqueue %.1f us/call, wall %.1f us/call" % (name, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): api()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
def t(fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (t1 - t0) / n * 1e6
print("construct only            %.1f us" % t(lambda: ts.DMV1o([d_, a_], lengths)))
print("partition (no grad inputs) %.1f us" % t(lambda: ts.DMV1o([md, ma], lengths).partition))
print("partition (grad inputs)   %.1f us" % t(lambda: ts.DMV1o([d_, a_], lengths).partition))
print("Fn.dmv1o_sum direct       %.1f us" % t(lambda: Fn.dmv1o_sum(d_, a_, lengths, 0)))
print("  + sum                   %.1f us" % t(lambda: Fn.dmv1o_sum(d_, a_, lengths, 0).sum()))
print("  + grad                  %.1f us" % t(lambda: torch.autograd.grad(Fn.dmv1o_sum(d_, a_, lengths, 0).sum(), [d_, a_])))
z = Fn.dmv1o_sum(d_, a_, lengths, 0)
gd, ga = Fn.dmv1o_run(md, ma, lengths, 0, True)[1:]
go = torch.ones(B, 1, device=dev).expand(B, 1)
print("_scale_counts alone       %.1f us" % t(lambda: Fn._scale_counts(gd, ga, go, torch.bfloat16, torch.bfloat16)))
print("stream_of                 %.1f us" % t(lambda: __import__('vlgae_amd')._C.stream_of(md)))
print("torch.empty x3            %.1f us" % t(lambda: (torch.empty(B, device=dev), torch.empty((B, 41, 2, 2, 2), device=dev), torch.empty((B, 41, 41, 2), device=dev))))
