"""attention-fuse forward+backward: vlgae_amd.align.attention_fuse (autograd -> vlg_attn_fuse_backward) vs the same lines
in torch ops (joint.py:670-674), config-2 shapes.  Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, L, V, d, h = 256, 40, 36, 128, 256
g = torch.Generator().manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    mk = lambda *s: torch.randn(*s, generator=g).to(dev, dt).requires_grad_(True)
    vis, txt, mid, enc = mk(B, V, d), mk(B, L + 1, d), mk(B, V, h), mk(B, L, h)
    ln = torch.nn.LayerNorm(h).to(dev)
    dout = torch.randn(B, L, h, generator=g).to(dev)
    leaves = [vis, txt, mid, enc, ln.weight, ln.bias]
    def ours():
        return torch.autograd.grad(align.attention_fuse(vis, txt, mid, enc, ln.weight, ln.bias, ln.eps), leaves, dout)
    def ref():
        att = torch.einsum("bvd,bqd->bqv", vis.float(), txt.float()[:, 1:]).softmax(2)
        return torch.autograd.grad(ln(enc.float() + torch.einsum("bqv,bvh->bqh", att, mid.float())), leaves, dout)
    err = max(float((a.float() - b.float()).abs().max()) for a, b in zip(ours(), ref()))
    print(dt, 'max grad err vs torch', err)
    for name, fn in (('vlg fwd+bwd', ours), ('torch fwd+bwd', ref)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f'  {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
