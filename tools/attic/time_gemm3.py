"""vlg_linear_wgrad at the parser feed-forwards' shapes (kernel + reduction, HIP events)."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
def ev(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
g = torch.Generator().manual_seed(0)
for K, M, N in ((41152, 256, 256), (41152, 32, 256), (20576, 512, 256), (20576, 256, 256), (10288, 512, 256), (10240, 256, 800), (10496, 384, 256)):
    dy = torch.randn(K, M, generator=g).to(dev).bfloat16(); x = torch.randn(K, N, generator=g).to(dev).bfloat16()
    dw, db = align.linear_wgrad(dy, x)
    ref = dy.double().t() @ x.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    print(f'K={K} M={M} N={N}: {ev(lambda: align.linear_wgrad(dy, x)):.1f} us  rel err {err:.1e}  {(K * (M + N) * 2) / 1e6:.0f} MB of operands', flush=True)
