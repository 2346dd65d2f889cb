import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B, L, V, d, h = 256, 40, 36, 128, 256
g = torch.Generator().manual_seed(0)
for dt in (torch.float32, torch.bfloat16):
    vis = torch.randn(B, V, d, generator=g).to(dev, dt); txt = torch.randn(B, L + 1, d, generator=g).to(dev, dt)
    mid = torch.randn(B, V, h, generator=g).to(dev, dt); enc = torch.randn(B, L, h, generator=g).to(dev, dt)
    ln = torch.nn.LayerNorm(h).to(dev)
    def ours(): return align.attention_fuse(vis, txt, mid, enc, ln.weight, ln.bias, ln.eps)
    def ref():
        att = torch.einsum("bvd,bqd->bqv", vis.float(), txt.float()[:, 1:]).softmax(2)
        return ln(enc.float() + torch.einsum("bqv,bvh->bqh", att, mid.float()))
    print(dt, 'max err', float((ours() - ref()).abs().max()))
    for name, fn in (('vlg_attn_fuse', ours), ('torch ops (3 kernels + LN)', ref)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        print(f'  {name}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us')
