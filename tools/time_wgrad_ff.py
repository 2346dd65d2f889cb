"""The split-K weight gradients of the training step's backward pass alone, at its shapes (B = 256, L = 40): python tools/time_wgrad_ff.py [substring]
-- graph-timed (20 calls per replay); the split-K launch alone (`partial`, what runs between the backward stages) and with its reduction."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import align
dev = torch.device('cuda:0')
bf = torch.bfloat16
def t(fn, n=20, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / (n * reps)
g = torch.Generator().manual_seed(0)
shapes = (("projections 4M0 x 32 x 256", 40960, 32, 256), ("linear1 / direction 4M x 256 x 256", 41152, 256, 256), ("left|right 2M x 512 x 256", 20576, 512, 256),
          ("valence 2M x 256 x 256", 20576, 256, 256), ("no|has M x 512 x 256", 10288, 512, 256), ("head M0 x 256 x 800", 10240, 256, 800),
          ("text encoder M0 x 256 x 800", 10240, 256, 800), ("visual encoder BR x 256 x 2048", 9216, 256, 2048), ("word|child|parent M0 x 768 x 256", 10496, 768, 256))
for name, K, M, N in shapes:
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    dy = torch.randn(K, M, generator=g).to(dev, bf); x = torch.randn(K, N, generator=g).to(dev, bf)
    out = (torch.empty(M, N, device=dev, dtype=bf), torch.empty(M, device=dev, dtype=bf))
    def partial():
        wg = align.WgradGroup()
        align.linear_wgrad(dy, x, out=out, defer=wg)
        partial.wg = wg
    def full():
        align.linear_wgrad(dy, x, out=out)
    tp, tf = t(partial), t(full)
    mb = (K * (M + N) * 2) / 1e6
    print("%-40s split-K launch %6.1f us (%5.2f TB/s of the operands' %5.1f MB), with its reduction %6.1f us" % (name, tp, mb / tp, mb, tf))
