import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgae_amd.torch_struct as ts
from vlgae_amd import align
dev = torch.device("cuda:0")
B, L, V, d, h = 256, 40, 36, 128, 256
g = torch.Generator().manual_seed(1)
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
attach = torch.randn(B, L, L, 2, generator=g).to(dev)
root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
d_, a_ = md.bfloat16().requires_grad_(True), ma.bfloat16().requires_grad_(True)
leaf = lambda *s: torch.randn(*s, generator=g).to(dev).bfloat16().requires_grad_(True)
vis, txt, mid, enc = leaf(B, V, d), leaf(B, L + 1, d), leaf(B, V, h), leaf(B, L, h)
lw, lb = torch.ones(h, device=dev, requires_grad=True), torch.zeros(h, device=dev, requires_grad=True)
def t(name, fn, n=1000):
    for _ in range(100): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-44s host %.1f us  wall %.1f us" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
api = lambda: torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
def fuse():
    out = align.attention_fuse(vis, txt, mid, enc, lw, lb, 1e-5)
    return torch.autograd.grad(out, [vis, txt, mid, enc, lw, lb], out)
for rep in range(2):
    t("api (engine thread)", api)
    t("attention_fuse fwd+bwd (engine thread)", fuse)
    with torch.autograd.set_multithreading_enabled(False):
        t("api (backward on the calling thread)", api)
        t("attention_fuse fwd+bwd (calling thread)", fuse)
