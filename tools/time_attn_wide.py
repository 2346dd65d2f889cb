"""attention-fuse at the shipped factor layout (V = 36 + 36^2 + 36 + 1 = 1369 keys, config/model/vlgae.yaml:40-42): forward + adjoint of
vlgae_amd.align.attention_fuse (key-split kernels, csrc/vlg_attn.hip) vs the same lines in torch ops (joint.py:670-674: batched library
GEMMs + softmax + LayerNorm + autograd).  HIP-graph replays so that the figures are device time, not Python enqueue.
    python tools/time_attn_wide.py [B] [key_chunk]
Run under rocprofv3 --kernel-trace --stats for per-kernel times."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
CK = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L, V, d, h = 40, 1369, 128, 256
g = torch.Generator().manual_seed(0)
for dt in (torch.bfloat16, torch.float32):
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.3).to(dev, dt).requires_grad_(True)
    vis, txt, mid, enc = mk(B, V, d), mk(B, L + 1, d), mk(B, V, h), mk(B, L, h)
    ln = torch.nn.LayerNorm(h).to(dev)
    dout = torch.randn(B, L, h, generator=g).to(dev)
    leaves = [vis, txt, mid, enc, ln.weight, ln.bias]
    def ours_fwd():
        with torch.no_grad():
            return align.attention_fuse(vis, txt, mid, enc, ln.weight, ln.bias, ln.eps, key_chunk=CK)
    def ours():
        return torch.autograd.grad(align.attention_fuse(vis, txt, mid, enc, ln.weight, ln.bias, ln.eps, key_chunk=CK), leaves, dout)
    def ref():
        f32 = torch.float32
        s = torch.bmm(txt[:, 1:].to(f32), vis.to(f32).transpose(1, 2))
        x = torch.bmm(torch.softmax(s, -1), mid.to(f32))
        return torch.autograd.grad(torch.nn.functional.layer_norm(enc.to(f32) + x, (h,), ln.weight, ln.bias, ln.eps), leaves, dout)
    err = max(float((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)) for a, b in zip(ours(), ref()))
    print(dt, f'B={B} key_chunk={CK}: max relative grad err vs torch {err:.2e}', flush=True)
    for name, fn in (('vlg fwd', ours_fwd), ('vlg fwd+bwd', ours), ('torch fwd+bwd', ref)):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): fn()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            keep = fn()
        for _ in range(5): graph.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): graph.replay()
        e1.record(); torch.cuda.synchronize()
        print(f'  {name}: {e0.elapsed_time(e1) / 30 * 1e3:.1f} us', flush=True)
