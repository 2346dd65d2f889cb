import sys, glob, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load, arcenc_w1
from vlgae_amd import langfeat
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for path in sorted(glob.glob('tests/golden/langfeat_*.npz')):
    g = load(path)
    x = t(g["x"]).requires_grad_(True)
    w_enc = np.concatenate([g["w_word"], g["w_child"], g["w_parent"]], 0); b_enc = np.concatenate([g["b_word"], g["b_child"], g["b_parent"]], 0)
    params = [t(a).requires_grad_(True) for a in (w_enc, b_enc, arcenc_w1(g), g["w2"], g["b_arc"])]
    txt, m, mg = langfeat.lang_feat_max_tree(x, t(g["lengths"]), t(g["merged_dec"]), t(g["merged_attach"]), *params, add_marginal=bool(g["add_marginal"]), slope=float(g["slope"]))
    print(path, 'txt err', float(np.abs(txt.detach().float().cpu().numpy() - g["txt"]).max()), 'max', float(np.abs(g["txt"]).max()))
    grads = torch.autograd.grad(txt, [x] + params, t(g["dout"]).to(txt.dtype))
    d = g["w2"].shape[0]
    got = {"x": grads[0], "w_word": grads[1][:d], "w_child": grads[1][d:2 * d], "w_parent": grads[1][2 * d:], "b_word": grads[2][:d], "b_child": grads[2][d:2 * d], "b_parent": grads[2][2 * d:], "w2": grads[4], "b_arc": grads[5]}
    for k, v in got.items():
        ref = g["g_" + k]; e = np.abs(v.float().cpu().numpy().astype(np.float64) - ref)
        print(f'  {k:9s} max err {e.max():.4g}  ref max {np.abs(ref).max():.4g}  rel L2 {np.linalg.norm(e) / np.linalg.norm(ref):.4g}')
    if "g_w1" in g:
        ref = g["g_w1"]; e = np.abs(grads[3].float().cpu().numpy() - ref); print(f'  w1 max err {e.max():.4g} ref max {np.abs(ref).max():.4g}')
    else:
        ref = g["g_w1_sample"]; e = np.abs(grads[3].float().cpu().numpy()[::5, ::7, ::3] - ref); print(f'  w1 sample max err {e.max():.4g} ref max {np.abs(ref).max():.4g}')
