"""a9 backward (both feature gradients of the materialised tensor), masked and unmasked call, bf16 features, config-2 shapes."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, Q, V, d = 256, 82, 36, 128
g = torch.Generator().manual_seed(0)
txt = torch.randn(B, Q, d, generator=g).to(dev).bfloat16(); vis = torch.randn(B, V, d, generator=g).to(dev).bfloat16()
cot = torch.randn(B, B, Q, V, generator=torch.Generator(device=dev).manual_seed(7), device=dev)
tm = torch.ones(B, Q, dtype=torch.bool, device=dev); tm[:, 0] = tm[:, 41] = False
vm = torch.ones(B, V, dtype=torch.bool, device=dev)
def ev(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print('unmasked %.3f ms   masked (root slots, joint.py:204) %.3f ms' % (ev(lambda: align.bilinear_align_backward(cot, txt, vis)),
                                                                       ev(lambda: align.bilinear_align_backward(cot, txt, vis, tm, vm))))
