import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B = A = 256; Q, V, d = 82, 36, 128
g = torch.Generator().manual_seed(5)
for dt in (torch.bfloat16,):
    txt = torch.randn(B, Q, d, generator=g).to(dev, dt); vis = torch.randn(A, V, d, generator=g).to(dev, dt)
    for kw in (dict(full=False, diag=True), dict(full=True), dict(full=False, max_v=True, max_q=True), dict(full=False, max_q=True), dict(full=True, max_v=True, max_q=True, diag=True)):
        for _ in range(5): r = align.bilinear_align(txt, vis, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): r = align.bilinear_align(txt, vis, **kw)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(dt, sorted(kw.items()), '%.3f ms' % ms, '%.0f GB/s out' % ((B*A*Q*V*4 if kw.get('full') else 0) / ms / 1e6), flush=True)
