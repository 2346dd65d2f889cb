"""align_max_kernel (fused maxima, no tensor) and its ARGS variant inside the grounding loss, config-2 shapes."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, Q, V, d = 256, 82, 36, 128
g = torch.Generator().manual_seed(0)
txt = torch.randn(B, Q, d, generator=g).to(dev).bfloat16(); vis = torch.randn(B, V, d, generator=g).to(dev).bfloat16()
tm = torch.ones(B, Q, dtype=torch.bool, device=dev); tm[:, 0] = tm[:, 41] = False
vm = torch.ones(B, V, dtype=torch.bool, device=dev)
marg = torch.rand(B, Q, generator=g).to(dev) * tm
def ev(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
print('fused maxima: %.3f ms' % ev(lambda: align.bilinear_align(txt, vis, tm, vm, full=False, max_v=True, max_q=True)))
with torch.no_grad():
    print('grounding loss forward (ARGS alignment + cross-entropies): %.3f ms' % ev(lambda: align.grounding_loss_factor_ce(txt, vis, tm, vm, marg, B * 40, 1.0)))
