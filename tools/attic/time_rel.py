"""VisBoxRelSimpleEncoder's rel path (box_rel.py:41-45): reference formulation in torch ops vs rel_features."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import vis_encoder
dev = torch.device('cuda:0')
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for B in (64, 256):
    for dt in (torch.float32, torch.bfloat16):
        R, n, H = 36, 2048, 256
        g = torch.Generator().manual_seed(1)
        feat = torch.randn(B, R, n, generator=g).to(dev, dt).requires_grad_(True)
        w = (torch.randn(H, 2 * n, generator=g) / (2 * n) ** 0.5).to(dev, dt).requires_grad_(True)
        b = torch.zeros(H, device=dev, dtype=dt, requires_grad=True)
        dout = torch.randn(B, R * R, H, generator=g).to(dev, dt)
        def ref(bwd):
            inputs = torch.cat([feat, feat.mean(1, keepdim=True).expand(-1, R, -1)], -1)
            rel = torch.nn.functional.leaky_relu(torch.nn.functional.linear((inputs.unsqueeze(1) + inputs.unsqueeze(2)) / 2, w, b)).view(B, R * R, H)
            return torch.autograd.grad(rel, [feat, w, b], dout) if bwd else rel
        def ours(bwd):
            rel = vis_encoder.rel_features(feat, w, b)
            return torch.autograd.grad(rel, [feat, w, b], dout) if bwd else rel
        fl = 2.0 * B * R * R * 2 * n * H
        print(f"B={B} {str(dt)[6:]:9s} reference formulation fwd {timeit(lambda: ref(False)):8.3f} ms ({fl/1e9:.0f} GFLOP) fwd+bwd {timeit(lambda: ref(True)):8.3f} ms | "
              f"rel_features fwd {timeit(lambda: ours(False)):7.3f} ms  fwd+bwd {timeit(lambda: ours(True)):7.3f} ms", flush=True)
