"""old vs new arg-max positions inside vlg_grounding_loss's workspace (GroundPlan offsets restated here)."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import _C
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
B, L, V, d = 6, 40, 36, 128
Q = 2 * (L + 1)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
txt = t((rng.standard_normal((B, Q, d)) * 0.5).astype(np.float32)).bfloat16()
vis = t((rng.standard_normal((B, V, d)) * 0.5).astype(np.float32)).bfloat16()
marg = t(rng.random((B, Q)).astype(np.float32))
pen = t((rng.integers(0, 3, (B, Q, 3)) * 100.0).astype(np.float32)); seg = t(rng.integers(0, 3, V).astype(np.uint8))
tm = torch.ones(B, Q, dtype=torch.uint8, device=dev); vm = torch.ones(B, V, dtype=torch.uint8, device=dev)
if len(sys.argv) > 1:
    tm[:, 0] = 0; tm[:, 41] = 0; vm[:, 5] = 0
up = lambda x: (x + 63) & ~63
nV, nQ = B * B * Q, B * B * V
off_maxQ = up(nV); off_part = off_maxQ + up(nQ)
def go(use_masks):
    nbytes = _C.lib().vlg_grounding_loss_workspace(B, Q, V)
    ws = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
    sums = torch.zeros(3, device=dev)
    _C.check(_C.lib().vlg_grounding_loss(_C.ptr(txt), _C.ptr(vis), _C.ptr(tm if use_masks else None), _C.ptr(vm if use_masks else None), _C.ptr(marg), _C.ptr(pen), _C.ptr(seg), 3, B, Q, V, d,
                                         _C.BF16, -1e20, 200.0, 1.0, _C.ptr(ws), nbytes, _C.ptr(sums), None, None, _C.stream_of(txt)), "g")
    torch.cuda.synchronize()
    raw = ws.cpu().numpy()
    # locate the uint16 arrays: off_coef = off_part + up(2*B*kCeMaxY) is unknown here -> search for argV by brute force is fragile;
    # use the documented order instead: argV follows coef (64 floats)
    return sums.cpu().numpy(), raw
for use_masks in (False, True):
    os.environ["VLG_ALIGN_ARGMAX_OLD"] = "1"
    s_old, w_old = go(use_masks)
    os.environ.pop("VLG_ALIGN_ARGMAX_OLD")
    s_new, w_new = go(use_masks)
    print('masks' if use_masks else 'none ', s_old, s_new)
    diff = np.nonzero(w_old != w_new)[0]
    print('  differing ws floats:', len(diff), diff[:20], ' (maxV < %d, maxQ < %d)' % (off_maxQ, off_part))
    for i in diff[:10]:
        if i < nV:
            b, r = divmod(i, B * Q); a, q = divmod(r, Q); print('   maxV b=%d a=%d q=%d old %g new %g' % (b, a, q, w_old[i], w_new[i]))
