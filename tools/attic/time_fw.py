import sys, torch
sys.path.insert(0, '.')
from vlgae_amd.torch_struct import functional as F
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(1)
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 40)
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev); attach = torch.randn(B, L, L, 2, generator=g).to(dev); root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
md, ma = md.bfloat16().contiguous(), ma.bfloat16().contiguous()   # headline storage type
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
def timeit(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print('B=%d L=%d  ' % (B, L) + 'inside-only %.1f us   fused %.1f us' % (timeit(lambda: F.dmv1o_run(md, ma, lengths, 0, False)), timeit(lambda: F.dmv1o_run(md, ma, lengths, 0, True))))
