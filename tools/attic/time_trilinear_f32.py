"""float32 arc trilinear (M = 10 496, X = H = Y = 128): the fp16-parts kernels against the exact-fp32 ones (VLG_TRI_F32_EXACT=1), event-timed,
with the error of each against float64.    python tools/time_trilinear_f32.py ; VLG_TRI_F32_EXACT=1 python tools/time_trilinear_f32.py"""
import os, sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
M = 10496
gen = torch.Generator().manual_seed(0)
child = (torch.randn(M, 128, generator=gen) * 0.5).to(dev).requires_grad_(True)
parent = (torch.randn(M, 128, generator=gen) * 0.5).to(dev).requires_grad_(True)
w1 = (torch.randn(128, 128, 128, generator=gen) / 128).to(dev).requires_grad_(True)
g = (torch.randn(M, 128, generator=gen) * 1e-3).to(dev)
def ev(fn, n=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
fwd = lambda: align.arc_trilinear(child, w1, parent)
def both():
    out = align.arc_trilinear(child, w1, parent)
    return torch.autograd.grad(out, [child, w1, parent], g)
t_f, t_b = ev(fwd), ev(both)
out = fwd(); grads = both()
c64, w64, p64 = (a.detach().double().requires_grad_(True) for a in (child, w1, parent))
t = torch.einsum("mhy,my->mh", torch.einsum("mx,xhy->mhy", c64, w64), p64)
refs = torch.autograd.grad(t, [c64, w64, p64], g.double())
errs = [float((out.double() - t.detach()).abs().max() / t.abs().max())] + [float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(grads, refs)]
print(f"{'exact fp32' if os.environ.get('VLG_TRI_F32_EXACT') else 'fp16 parts'}: forward {t_f:.1f} us, forward + backward {t_b:.1f} us; max err / max |ref|: out {errs[0]:.1e}, d_child {errs[1]:.1e}, d_w {errs[2]:.1e}, d_parent {errs[3]:.1e}")
