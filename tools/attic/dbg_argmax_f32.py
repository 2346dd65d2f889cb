"""float32 grounding loss at config-2 widths: arg-max positions of the fp16-parts kernel (default) and of the exact-fp32 MFMA kernel
(VLG_ALIGN_F32_EXACT=1, a child process) against float64 -- how many positions differ, and how close the two candidates were.
    python tools/dbg_argmax_f32.py [B]"""
import os, sys, json, subprocess, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import _C
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L, V, d = 40, 36, 128
Q = 2 * (L + 1)
KCEMAXY = 8

def run():
    g = torch.Generator().manual_seed(3)
    txt = (torch.randn(B, Q, d, generator=g) * 0.5).to(dev); vis = (torch.randn(B, V, d, generator=g) * 0.5).to(dev)
    lengths = torch.randint(L // 2, L + 1, (B,), generator=g)
    m1 = torch.cat([torch.zeros(B, 1, dtype=torch.bool), torch.arange(L)[None] < lengths[:, None]], 1)
    tmask = torch.cat([m1, m1], 1).to(dev); vmask = (torch.rand(B, V, generator=g) > 0.2).to(dev); vmask[:, 0] = True
    marg = (torch.rand(B, Q, generator=g).to(dev) * tmask)
    lib = _C.lib()
    nbytes = lib.vlg_grounding_loss_workspace(B, Q, V)
    ws = torch.zeros(nbytes // 4, device=dev); sums = torch.zeros(3, device=dev)
    tm, vm = tmask.to(torch.uint8), vmask.to(torch.uint8)
    _C.check(lib.vlg_grounding_loss(_C.ptr(txt), _C.ptr(vis), _C.ptr(tm), _C.ptr(vm), _C.ptr(marg), None, None, 0, B, Q, V, d, _C.F32, -1e20,
                                    float(lengths.sum()), 1.0, _C.ptr(ws), nbytes, _C.ptr(sums), None, None, None), "grounding_loss")
    torch.cuda.synchronize()
    up = lambda x: (x + 63) & ~63
    nV, nQ = B * B * Q, B * B * V
    off_argV = up(nV) + up(nQ) + up(2 * B * KCEMAXY) + 64
    off_argQ = off_argV + up((nV + 1) // 2)
    argV = ws[off_argV:off_argV + (nV + 1) // 2].view(torch.int16)[:nV].view(B, B, Q).to(torch.int64) & 0xffff
    argQ = ws[off_argQ:off_argQ + (nQ + 1) // 2].view(torch.int16)[:nQ].view(B, B, V).to(torch.int64) & 0xffff
    # float64 scores
    S = torch.einsum("bqd,avd->baqv", txt.double(), vis.double())
    S = S.masked_fill(~vmask[None, :, None, :], -1e20).masked_fill(~tmask[:, None, :, None], -1e20)
    out = {}
    for name, arg, dim in (("over regions", argV, 3), ("over queries", argQ, 2)):
        top2 = S.topk(2, dim=dim).values
        best = S.argmax(dim)   # (first position on ties in torch >= 1.7)
        gap = (top2.select(dim, 0) - top2.select(dim, 1)) / top2.select(dim, 0).abs().clamp_min(1e-30)
        live = top2.select(dim, 0) > -1e19
        diff = (arg != best) & live
        picked = S.gather(dim, arg.unsqueeze(dim)).squeeze(dim)
        loss = ((top2.select(dim, 0) - picked) / top2.select(dim, 0).abs().clamp_min(1e-30))[diff]
        out[name] = dict(rows=int(live.sum()), differ=int(diff.sum()), worst_relative_shortfall=float(loss.max()) if diff.any() else 0.0,
                         rows_with_gap_below_1e_6=int(((gap < 1e-6) & live).sum()))
    out["sums"] = sums.tolist()
    return out

if os.environ.get("VLG_DBG_CHILD"):
    print(json.dumps(run())); sys.exit(0)
for exact in ("", "1"):
    env = dict(os.environ, VLG_DBG_CHILD="1")
    env.pop("VLG_ALIGN_F32_EXACT", None)
    if exact: env["VLG_ALIGN_F32_EXACT"] = "1"
    r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
    if r.returncode: print(r.stderr[-2000:]); sys.exit(1)
    print("exact-fp32 MFMA kernel:" if exact else "fp16-parts kernel:", r.stdout.strip().splitlines()[-1])
