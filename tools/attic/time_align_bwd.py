"""a9 backward: gradients of gather_logit_simple's materialised [B,A,Q,V] tensor w.r.t. both feature tensors."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, Q, V, d = 256, 82, 36, 128
g = torch.Generator().manual_seed(5)
for dt in (torch.bfloat16, torch.float32):
    txt = torch.randn(B, Q, d, generator=g).to(dev, dt).requires_grad_(True)
    vis = torch.randn(B, V, d, generator=g).to(dev, dt).requires_grad_(True)
    tm = torch.ones(B, Q, dtype=torch.bool, device=dev); tm[:, 0] = tm[:, 41] = False
    vm = torch.rand(B, V, generator=g).to(dev) > 0.1
    cot = torch.randn(B, B, Q, V, generator=g).to(dev)
    out = align._GatherLogit.apply(txt, vis, tm, vm, -1e20)
    def bwd():
        return torch.autograd.grad(out, [txt, vis], cot, retain_graph=True)
    for _ in range(3): bwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): bwd()
    e1.record(); torch.cuda.synchronize()
    print(dt, 'backward of the materialised tensor: %.3f ms' % (e0.elapsed_time(e1) / 10))
    del out, cot

# per side, raw (no autograd): one kernel each
txt = torch.randn(B, Q, d, generator=g).to(dev, torch.bfloat16); vis = torch.randn(B, V, d, generator=g).to(dev, torch.bfloat16)
cot = torch.randn(B, B, Q, V, generator=g).to(dev)
tm = torch.ones(B, Q, dtype=torch.bool, device=dev); vm = torch.ones(B, V, dtype=torch.bool, device=dev)
for name, kw in (("d_txt", dict(want_vis=False)), ("d_vis", dict(want_txt=False))):
    for masks in ((None, None), (tm, vm)):
        fn = lambda: align.bilinear_align_backward(cot, txt, vis, masks[0], masks[1], **kw)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        print(name, 'masks' if masks[0] is not None else 'no masks', '%.3f ms' % (e0.elapsed_time(e1) / 10))
