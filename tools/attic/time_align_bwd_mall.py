"""Does the a9 backward run faster when its cotangent is Infinity-Cache resident?  Per side (one kernel each), per (B = A) size: the
cotangent [B,B,82,36] fp32 is 774 MB at B = 256 (streams from HBM), 193 MB at B = 128 and 109 MB at B = 96 (stay in the 256 MB cache
between back-to-back calls).  If the per-byte time does not drop for the resident sizes, serving the second reader of every block from
the cache (the skewed two-sided launch of DESIGN section 7 item 4) cannot pay."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
Q, V, d = 82, 36, 128
g = torch.Generator().manual_seed(5)
for B in (256, 192, 128, 96, 64):
    txt = torch.randn(B, Q, d, generator=g).to(dev, torch.bfloat16); vis = torch.randn(B, V, d, generator=g).to(dev, torch.bfloat16)
    cot = torch.randn(B, B, Q, V, generator=g).to(dev)
    mb = cot.numel() * 4 / 1e6
    res = []
    for name, kw in (("d_txt", dict(want_vis=False)), ("d_vis", dict(want_txt=False)), ("both", dict())):
        fn = lambda: align.bilinear_align_backward(cot, txt, vis, None, None, **kw)
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res.append(f"{name} {ms * 1e3:7.1f} us = {mb * (2 if name == 'both' else 1) / ms / 1e3:5.2f} TB/s")
    print(f"B=A={B:3d} cotangent {mb:6.1f} MB: " + " | ".join(res), flush=True)
