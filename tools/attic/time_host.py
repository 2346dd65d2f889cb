"""Host enqueue time vs GPU time of the Python-level entry points (is an op host-bound inside a training step?)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgae_amd.torch_struct as ts
from vlgae_amd import align

dev = torch.device("cuda:0")
torch.manual_seed(0)


def measure(name, fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print(f"{name:34s} host enqueue {t_host * 1e6:8.1f} us   wall {t_all * 1e6:8.1f} us", flush=True)


B, L, V, d, h = 256, 40, 36, 128, 256
N = L + 1
dec = torch.randn(B, N, 2, 2, 2, device=dev).log_softmax(-1).requires_grad_()
att = torch.randn(B, N, N, 2, device=dev).requires_grad_()
lengths = torch.full((B,), L, device=dev)


def dp():
    torch.autograd.grad(ts.DMV1o([dec, att], lengths).partition.sum(), [dec, att])


measure("DMV1o partition + grad", dp)
measure("DMV1o marginals_and_heads", lambda: ts.DMV1o([dec, att], lengths).marginals_and_heads())

vis = torch.randn(B, V, d, device=dev, dtype=torch.bfloat16).requires_grad_()
txt = torch.randn(B, L + 1, d, device=dev, dtype=torch.bfloat16).requires_grad_()
mid = torch.randn(B, V, h, device=dev, dtype=torch.bfloat16).requires_grad_()
x = torch.randn(B, L, h, device=dev, dtype=torch.bfloat16).requires_grad_()
gm = torch.ones(h, device=dev).requires_grad_()
bt = torch.zeros(h, device=dev).requires_grad_()
dout = torch.randn(B, L, h, device=dev, dtype=torch.bfloat16)


def fuse():
    y = align.attention_fuse(vis, txt, mid, x, gm, bt, 1e-5)
    torch.autograd.grad(y, [vis, txt, mid, x, gm, bt], dout.to(y.dtype))


measure("attention_fuse fwd + bwd", fuse)

M = B * N
c = (torch.randn(M, 128, device=dev, dtype=torch.bfloat16) * 0.3).requires_grad_()
p = (torch.randn(M, 128, device=dev, dtype=torch.bfloat16) * 0.3).requires_grad_()
w = (torch.randn(128, 128, 128, device=dev, dtype=torch.bfloat16) * 0.1).requires_grad_()
g = torch.randn(M, 128, device=dev)


def tri():
    y = align.arc_trilinear(c, w, p)
    torch.autograd.grad(y, [c, w, p], g.to(y.dtype))


measure("arc_trilinear fwd + bwd", tri)

if len(sys.argv) > 1 and sys.argv[1] == "--profile":
    import cProfile
    import pstats
    for name, fn in (("dp", dp), ("fuse", fuse)):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(300):
            fn()
        pr.disable()
        torch.cuda.synchronize()
        print("=====", name)
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
