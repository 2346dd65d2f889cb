import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgae_amd.torch_struct as ts
from vlgae_amd.torch_struct import functional as Fn
from vlgae_amd.torch_struct.semirings import LogSemiring
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
def t(name, fn, n=300):
    for _ in range(30): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%-44s host %.1f us  wall %.1f us" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
t("A dmv1o_sum direct", lambda: Fn.dmv1o_sum(d_, a_, lengths, 0))
t("B DMV1o.partition", lambda: ts.DMV1o([d_, a_], lengths).partition)
def C():
    x = ts.DMV1o([d_, a_], lengths); return x._sum(LogSemiring)
t("C DMV1o._sum", C)
def D():
    with torch.enable_grad(): return Fn.dmv1o_sum(d_, a_, lengths, 0)
t("D enable_grad + dmv1o_sum", D)
t("A again", lambda: Fn.dmv1o_sum(d_, a_, lengths, 0))
t("A n=1000", lambda: Fn.dmv1o_sum(d_, a_, lengths, 0), 1000)
t("raw n=1000", lambda: Fn.dmv1o_run(md, ma, lengths, 0, True), 1000)
t("api n=300", lambda: torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_]))
t("api n=1000", lambda: torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_]), 1000)
fn = lambda: torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
torch.cuda.synchronize()
for chunk in range(16):
    t0 = time.perf_counter()
    for _ in range(100): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("chunk %2d: host %.1f us  wall %.1f us" % (chunk, (t1 - t0) / 100 * 1e6, (t2 - t0) / 100 * 1e6), flush=True)
import gc
print("gc counts", gc.get_count(), gc.get_stats()[-1])
