"""Runs every hot kernel of the package a few times at config-2 shapes, under the names the bench entries launch them by
(for tools/prof_kernels.sh: rocprofv3 kernel-trace and PMC passes).  bf16 features unless --f32."""
import sys
import torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from vlgae_amd import train_step
from vlgae_amd import align
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
dt = torch.float32 if '--f32' in sys.argv else torch.bfloat16
n = 3
B, L, V, d, h = 256, 40, 36, 128, 256
N, Q = L + 1, 2 * (L + 1)
g = torch.Generator().manual_seed(5)
mk = lambda *s: torch.randn(*s, generator=g).to(dev, dt).requires_grad_(True)
with torch.autograd.set_multithreading_enabled(False):
    # the training step in the reference's wiring (align_argmax, ground_bwd_ws, ground_ce, tri2 / tri_dw2, gemm_tn, scorer, attn_fuse, langfeat, DP x2)
    step = train_step.build(B, L, V, dev, dtype=dt)
    for _ in range(n):
        step()
    # the alignment entry points of the bench line
    txt, vis = mk(B, Q, d), mk(B, V, d)
    tmask = torch.ones(B, Q, dtype=torch.bool, device=dev); tmask[:, 0] = False; tmask[:, N] = False
    vmask = torch.rand(B, V, generator=g).to(dev) > 0.1
    for _ in range(n):
        align.bilinear_align(txt, vis, full=False, max_v=True, max_q=True)                # align_max_kernel
        full = align.bilinear_align(txt, vis, full=True)["full"]                           # align_full_kernel
        gout = torch.ones_like(full)
        align.bilinear_align_backward(gout, txt, vis, tmask, vmask)                        # align_bwd_split*_kernel
        del full, gout
    # DP headline and its siblings
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
    attach, root = torch.randn(B, L, L, 2, generator=g).to(dev), torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
    md, ma = ts.DMV1o.merge(dec, attach, root)
    md, ma = md.to(torch.bfloat16).requires_grad_(True), ma.to(torch.bfloat16).requires_grad_(True)
    lengths = torch.full((B,), L, dtype=torch.long, device=dev)
    for _ in range(n):
        torch.autograd.grad(ts.DMV1o([md, ma], lengths).partition.sum(), [md, ma])
torch.cuda.synchronize()
