"""Viterbi decode (Max inside + back-pointer walk): dmv1o_decode / deptree_decode at several (B, L)."""
import sys, torch
sys.path.insert(0, ".")
import vlgae_amd.torch_struct as ts
from vlgae_amd.torch_struct import functional as F
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for B, L in ((256, 40), (1024, 40), (4096, 40), (256, 80)):
    g = torch.Generator().manual_seed(1)
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev); attach = torch.randn(B, L, L, 2, generator=g).to(dev); root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
    md, ma = ts.DMV1o.merge(dec, attach, root)
    lengths = torch.full((B,), L, dtype=torch.long, device=dev)
    arc = ma[..., 0].contiguous()
    print("B=%d L=%d: dmv1o_decode %.1f us  (Max inside only %.1f, inside + replay with counts %.1f)   deptree_decode %.1f us" % (
        B, L, timeit(lambda: F.dmv1o_decode(md, ma, lengths)), timeit(lambda: F.dmv1o_run(md, ma, lengths, 1, False)),
        timeit(lambda: F.dmv1o_run(md, ma, lengths, 1, True)), timeit(lambda: F.deptree_decode(arc, lengths))))
B, L = 256, 40
g = torch.Generator().manual_seed(1)
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev); attach = torch.randn(B, L, L, 2, generator=g).to(dev); root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
d = lambda: ts.DMV1o([md, ma], lengths)
print("B=256 L=40: marginals then argmax_heads (one stream) %.1f us   marginals_and_heads (two streams) %.1f us" % (
    timeit(lambda: (d().marginals, d().argmax_heads)), timeit(lambda: d().marginals_and_heads())))
