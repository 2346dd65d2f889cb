"""configs[3] (B = 256, L = 80) of the library VLGAE_AMD_LIB points at: fused / inside-only time and a checksum."""
import sys, torch, hashlib
sys.path.insert(0, '.')
from vlgae_amd.torch_struct import functional as F
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
B, L = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 80
g = torch.Generator().manual_seed(1)
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
attach = torch.randn(B, L, L, 2, generator=g).to(dev)
root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
md, ma = md.bfloat16(), ma.bfloat16()
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
out = F.dmv1o_run(md, ma, lengths, 0, True)
torch.cuda.synchronize()
digest = hashlib.sha256(b''.join(x.detach().cpu().numpy().tobytes() for x in out if x is not None)).hexdigest()[:16]
print('L=%d fused %.1f us   inside %.1f us   sha256 %s' % (L, t(lambda: F.dmv1o_run(md, ma, lengths, 0, True)), t(lambda: F.dmv1o_run(md, ma, lengths, 0, False)), digest))
