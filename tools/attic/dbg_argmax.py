"""A-B of align_argmax_kernel vs align_max_kernel<true,3> inside the grounding loss (sums only), small masked shapes."""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
def run(B, L, V, with_pen, masked, seed=0):
    rng = np.random.default_rng(seed)
    Q = 2 * (L + 1); d = 128
    lengths = rng.integers(max(1, L // 2), L + 1, B)
    m1 = np.concatenate([np.zeros((B, 1), bool), np.arange(L)[None] < lengths[:, None]], 1)
    tmask = np.concatenate([m1, m1], 1) if masked else np.ones((B, Q), bool)
    vmask = rng.random((B, V)) > 0.2 if masked else np.ones((B, V), bool)
    vmask[:, 0] = True
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    txt = t((rng.standard_normal((B, Q, d)) * 0.5).astype(np.float32)).bfloat16()
    vis = t((rng.standard_normal((B, V, d)) * 0.5).astype(np.float32)).bfloat16()
    marg = t((rng.random((B, Q)) * tmask).astype(np.float32))
    pen = seg = None
    if with_pen:
        pen = t((rng.integers(0, 3, (B, Q, 3)) * 100.0).astype(np.float32)); seg = t(rng.integers(0, 3, V).astype(np.uint8))
    with torch.no_grad():
        total, sums = align.grounding_loss_factor_ce(txt, vis, t(tmask), t(vmask), marg, int(lengths.sum()), 1.0, pen, seg)
    return sums.cpu().numpy().tolist()


SHAPES = ((6, 40, 36), (8, 40, 36), (16, 20, 36), (6, 40, 32))
CASES = [(B, L, V, wp, mk) for B, L, V in SHAPES for wp in (False, True) for mk in (False, True)]
if os.environ.get("VLG_DBG_ARGMAX_CHILD"):
    # one kernel variant per PROCESS: the VLG_* switches are read once per process (VLG_ENV, csrc/vlg_common.h), so toggling
    # os.environ between two calls of one process compares a kernel with itself
    import json
    print(json.dumps([run(*c) for c in CASES]))
    sys.exit(0)
import json, subprocess
res = {}
for old in ("1", ""):
    env = dict(os.environ, VLG_DBG_ARGMAX_CHILD="1")
    env.pop("VLG_ALIGN_ARGMAX_OLD", None)
    if old:
        env["VLG_ALIGN_ARGMAX_OLD"] = "1"
    res[old] = json.loads(subprocess.run([sys.executable, __file__], env=env, check=True, capture_output=True, text=True).stdout.strip().splitlines()[-1])
for c, a, b in zip(CASES, res["1"], res[""]):
    print(*c, a, b, 'OK' if np.allclose(a, b, rtol=1e-5) else 'DIFF')
