"""Materialised alignment tensor [256,256,82,36] fp32 from bf16 features: new direct-store kernel vs the LDS-tile kernel, and torch fill_ of the same buffer."""
import os, sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, Q, V, d = 256, 82, 36, 128
g = torch.Generator().manual_seed(0)
txt = torch.randn(B, Q, d, generator=g).to(dev).bfloat16(); vis = torch.randn(B, V, d, generator=g).to(dev).bfloat16()
tm = torch.ones(B, Q, dtype=torch.bool, device=dev); tm[:, 0] = tm[:, 41] = False
vm = (torch.rand(B, V, generator=g) > 0.1).to(dev)
def ev(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for masks in (False, True):
    kw = dict(txt_mask=tm, vis_mask=vm) if masks else {}
    new = align.bilinear_align(txt, vis, **kw)["full"]
    both = align.bilinear_align(txt, vis, max_v=True, **kw)       # takes the LDS-tile kernel
    assert os.environ.get("VLG_SKIP_CHECK") or torch.equal(new, both["full"]), float((new - both["full"]).abs().max())   # (ablation builds: wrong by construction)
    t_new = ev(lambda: align.bilinear_align(txt, vis, **kw))
    t_old = ev(lambda: align.bilinear_align(txt, vis, max_v=True, **kw))
    nbytes = new.numel() * 4
    print(f'masks={masks}: direct-store {t_new:.3f} ms = {nbytes / t_new / 1e9:.2f} TB/s; LDS-tile (+max_v) {t_old:.3f} ms', flush=True)
buf = torch.empty(B, B, Q, V, device=dev)
print(f'fill_: {ev(lambda: buf.fill_(1.0)):.3f} ms')
