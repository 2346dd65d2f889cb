import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, Q, V, d = 256, 82, 36, 128
g = torch.Generator().manual_seed(5)
txt = torch.randn(B, Q, d, generator=g).to(dev, torch.bfloat16)
vis = torch.randn(B, V, d, generator=g).to(dev, torch.bfloat16)
tm = torch.ones(B, Q, dtype=torch.bool, device=dev); tm[:, 0] = tm[:, 41] = False
def timeit(fn, n=100):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
fl = 2.0 * B * B * Q * V * d
for name, kw in (("full", dict(full=True)), ("max_v+max_q", dict(full=False, max_v=True, max_q=True)), ("max_q", dict(full=False, max_q=True)),
                 ("max_v", dict(full=False, max_v=True)), ("max_v+max_q masked", dict(full=False, max_v=True, max_q=True, txt_mask=tm)),
                 ("max_v+diag", dict(full=False, max_v=True, diag=True))):
    ms = timeit(lambda: align.bilinear_align(txt, vis, **kw))
    print(f"{name:22s} {ms:.4f} ms  {fl / ms / 1e9:.0f} TFLOP/s")
# correctness of the fused maxima against reductions of the full tensor
vm = torch.rand(B, V, generator=g).to(dev) > 0.1
for masks in ((None, None), (tm, None), (tm, vm)):
    r = align.bilinear_align(txt, vis, masks[0], masks[1], full=True, max_v=True, max_q=True)
    r2 = align.bilinear_align(txt, vis, masks[0], masks[1], full=False, max_v=True, max_q=True)
    print("bit-equal:", torch.equal(r2["max_v"], r["full"].max(3).values), torch.equal(r2["max_q"], r["full"].max(2).values),
          torch.equal(r["max_v"], r2["max_v"]))
