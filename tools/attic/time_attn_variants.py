"""Times vlg_attn_fuse for each diagnosis build in tools/_v/attn_<phases>.so (children, one library each)."""
import os, subprocess, sys, glob
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, '.')
    from vlgae_amd import align
    dev = torch.device('cuda:0')
    B, L, V, d, h = 256, 40, 36, 128, 256
    g = torch.Generator().manual_seed(0)
    vis = torch.randn(B, V, d, generator=g).to(dev); txt = torch.randn(B, L + 1, d, generator=g).to(dev)
    mid = torch.randn(B, V, h, generator=g).to(dev); enc = torch.randn(B, L, h, generator=g).to(dev)
    w = torch.ones(h, device=dev); bb = torch.zeros(h, device=dev)
    fn = lambda: align.attention_fuse(vis, txt, mid, enc, w, bb, 1e-5)
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{sys.argv[1]}: {e0.elapsed_time(e1) / 200 * 1e3:.1f} us', flush=True)
else:
    for so in sorted(glob.glob('tools/_v/attn_*.so')):
        subprocess.run([sys.executable, __file__, os.path.basename(so)], env=dict(os.environ, VLGAE_AMD_LIB=os.path.abspath(so)))
