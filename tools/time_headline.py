"""Headline launch (B = 256, L = 40, bf16 potentials, Log, fused inside + outside) of the library VLGAE_AMD_LIB points at."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd.torch_struct import functional as F
import vlgae_amd.torch_struct as ts
dev = torch.device('cuda:0')
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 40)
g = torch.Generator().manual_seed(1)
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
attach = torch.randn(B, L, L, 2, generator=g).to(dev)
root = torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root)
md, ma = md.bfloat16(), ma.bfloat16()
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / n
out = F.dmv1o_run(md, ma, lengths, 0, True)
import hashlib
torch.cuda.synchronize()
digest = hashlib.sha256(b''.join(x.detach().cpu().numpy().tobytes() for x in out if x is not None)).hexdigest()[:16]
print('fused %.2f us   inside %.2f us   logZ[0] %.6f  sum|g| %.6f  sha256(logZ|counts) %s' % (t(lambda: F.dmv1o_run(md, ma, lengths, 0, True)), t(lambda: F.dmv1o_run(md, ma, lengths, 0, False)),
      float(out[0][0]), float(sum(x.float().abs().sum() for x in out[1:] if x is not None)), digest))
