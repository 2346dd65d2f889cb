"""Arc-encoder trilinear term (joint.py:282-284) at config-2: M = 256 x 41 rows, 128^3 weights; forward and forward +
gradients, vlgae_amd.align.arc_trilinear vs torch.einsum (which materialises [M,H,Y]).  Run under rocprofv3 for kernels."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
B, C, X = 256, 41, 128
for dt in (torch.bfloat16, torch.float32):
    mk = lambda *s, sc=0.5: (torch.randn(*s, generator=g) * sc).to(dev, dt).requires_grad_(True)
    child, parent, w1 = mk(B, C, X), mk(B, C, X), mk(X, X, X, sc=1.0 / X)
    dout = torch.randn(B, C, X, generator=g).to(dev)
    leaves = [child, w1, parent]
    ours_f = lambda: align.arc_trilinear(child.detach(), w1.detach(), parent.detach())
    ref_f = lambda: torch.einsum('bcx,xhy,bcy->bch', child.detach(), w1.detach(), parent.detach())
    ours_fb = lambda: torch.autograd.grad(align.arc_trilinear(child, w1, parent), leaves, dout)
    ref_fb = lambda: torch.autograd.grad(torch.einsum('bcx,xhy,bcy->bch', child, w1, parent), leaves, dout.to(dt))
    ref64 = torch.einsum('bcx,xhy,bcy->bch', child.double(), w1.double(), parent.double())
    print(dt, 'fwd err ours', float((ours_f() - ref64).abs().max()), 'torch', float((ref_f().double() - ref64).abs().max()))
    ga, gb = ours_fb(), ref_fb()
    print('   grad diff vs torch', [float((a.float() - b.float()).abs().max() / b.float().abs().max()) for a, b in zip(ga, gb)])
    for name, fn in (('ours fwd', ours_f), ('torch fwd', ref_f), ('ours fwd+bwd', ours_fb), ('torch fwd+bwd', ref_fb)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(f'   {name}: {e0.elapsed_time(e1) / 10 * 1e3:.1f} us')
