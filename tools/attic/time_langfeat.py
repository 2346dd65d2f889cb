"""lang_feat_max_tree forward + backward at config size: eager ms and kernel list (run under tools/prof_any.sh)."""
import sys, time, torch
sys.path.insert(0, '.')
import vlgae_amd.torch_struct as ts
from vlgae_amd import langfeat
dev = torch.device('cuda:0'); bf = torch.bfloat16
B, L, h, d = 256, 40, 256, 128
g = torch.Generator().manual_seed(5)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
x = rnd(B, L, h, sc=0.5).requires_grad_(True)
params = [rnd(3 * d, h, sc=h ** -0.5).to(bf).requires_grad_(True), rnd(3 * d, sc=0.1).to(bf).requires_grad_(True),
          rnd(d, d, d, sc=1.0 / d).to(bf).requires_grad_(True), rnd(d, d, sc=d ** -0.5).to(bf).requires_grad_(True), rnd(d, sc=0.1).to(bf).requires_grad_(True)]
dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
attach, root = torch.randn(B, L, L, 2, generator=g).to(dev), torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
md, ma = ts.DMV1o.merge(dec, attach, root); md, ma = md.to(bf), ma.to(bf)
lengths = torch.full((B,), L, dtype=torch.long, device=dev)
dout = rnd(B, 2 * (L + 1), d).to(bf)
def step():
    txt, m, mg = langfeat.lang_feat_max_tree(x, lengths, md, ma, *params, keep_viterbi=True)
    return torch.autograd.grad(txt, [x] + params, dout)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(30): step()
torch.cuda.synchronize(); print('lang_feat_max_tree fwd+bwd: %.3f ms' % ((time.perf_counter() - t0) / 30 * 1e3))
