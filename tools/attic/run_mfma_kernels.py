"""Runs every matrix-core kernel of the package a few times at config-2 shapes (for tools/prof_mfma.sh PMC passes)."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, L, V, d, h = 256, 40, 36, 128, 256
N, Q = L + 1, 2 * (L + 1)
g = torch.Generator().manual_seed(5)
mk = lambda *s: torch.randn(*s, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
txt, vis = mk(B, Q, d), mk(B, V, d)
f_vis, f_txt, f_mid, f_enc = mk(B, V, d), mk(B, N, d), mk(B, V, h), mk(B, L, h)
ln_w, ln_b = torch.ones(h, device=dev, requires_grad=True), torch.zeros(h, device=dev, requires_grad=True)
dout = torch.randn(B, L, h, generator=g).to(dev)
a_child, a_parent = mk(B, N, d), mk(B, N, d)
a_w1 = (torch.randn(d, d, d, generator=g) / d).to(dev, torch.bfloat16).requires_grad_(True)
a_dout = torch.randn(B, N, d, generator=g).to(dev)
leaves = [f_vis, f_txt, f_mid, f_enc, ln_w, ln_b]
for _ in range(6):
    align.bilinear_align(txt, vis, full=False, max_v=True, max_q=True)
    align.bilinear_align(txt, vis, full=True)
    torch.autograd.grad(align.attention_fuse(*leaves, 1e-5), leaves, dout)
    torch.autograd.grad(align.arc_trilinear(a_child, a_w1, a_parent), [a_child, a_w1, a_parent], a_dout)
torch.cuda.synchronize()
