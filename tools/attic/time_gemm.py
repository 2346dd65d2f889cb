"""Timing + check of vlg_linear_wgrad against torch's own GEMM on the training step's shapes."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align, _C
dev = torch.device('cuda:0')
def ev(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
g = torch.Generator().manual_seed(0)
for K, M, N in ((10496, 128, 256), (10496, 128, 128), (10496, 384, 256), (10496, 256, 256), (10496, 64, 64), (4096, 128, 2048), (9216, 256, 2048), (3000, 64, 192)):
    dy = torch.randn(K, M, generator=g).to(dev).bfloat16(); x = torch.randn(K, N, generator=g).to(dev).bfloat16()
    dw, db = align.linear_wgrad(dy, x)
    ref = dy.double().t() @ x.double(); refb = dy.double().sum(0)
    e1 = float((dw.double() - ref).abs().max() / ref.abs().max()); e2 = float((db.double() - refb).abs().max() / refb.abs().max())
    dw2, _ = align.linear_wgrad(dy, x)
    same = bool(torch.equal(dw, dw2))
    t_ours = ev(lambda: align.linear_wgrad(dy, x))
    t_lib = ev(lambda: dy.t() @ x)
    t_lib2 = ev(lambda: dy.float().sum(0))
    print(f'K={K} M={M} N={N}: ours {t_ours:.1f} us  torch matmul {t_lib:.1f} us (+ bias sum {t_lib2:.1f})  rel err dW {e1:.2e} db {e2:.2e}  reproducible {same}', flush=True)
# strided views (column slices of a wider buffer)
K = 10496
wide = torch.randn(K, 384, generator=g).to(dev).bfloat16(); x = torch.randn(K, 256, generator=g).to(dev).bfloat16()
dw, db = align.linear_wgrad(wide[:, 128:256], x)
ref = wide[:, 128:256].double().t() @ x.double()
print('strided slice rel err', float((dw.double() - ref).abs().max() / ref.abs().max()))
