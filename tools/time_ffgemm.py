"""The fused row-streaming Linear + element-wise pass (vlg_ff_linear_act / _backward) alone, at the training step's row counts:
python tools/time_ffgemm.py [rows] [only: a substring of the variant name]  -- graph-timed, with the library GEMM + vlg_ff_act pair beside it."""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd import parser_ff
dev = torch.device('cuda:0')
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 41152
H, bf = 256, torch.bfloat16
g = torch.Generator().manual_seed(0)
x = (torch.randn(rows, H, generator=g) * 0.5).to(dev, bf)
w = (torch.randn(H, H, generator=g) / 16).to(dev, bf)
w2 = (torch.randn(2 * H, H, generator=g) / 16).to(dev, bf)
b, b2 = torch.zeros(H, device=dev, dtype=bf), torch.zeros(2 * H, device=dev, dtype=bf)
out, out2 = torch.empty(rows, H, device=dev, dtype=bf), torch.empty(2 * rows, H, device=dev, dtype=bf)
act = torch.randn(rows, H, generator=g).to(dev, bf)
tot = torch.zeros(rows // 4, H, device=dev)
x512 = (torch.randn(rows, 2 * H, generator=g) * 0.5).to(dev, bf)
x32 = (torch.randn(rows, 32, generator=g) * 0.5).to(dev, bf)
w32 = (torch.randn(32, H, generator=g) / 6).to(dev, bf)
w512 = (torch.randn(2, H, H, generator=g) / 22).to(dev, bf)            # the two transposed blocks
w512_kn = torch.cat([w512[0].t(), w512[1].t()], 0).contiguous()        # [512, 256]
def t(fn, n=20, reps=10):
    """GPU time per call: n calls captured as one HIP graph (eager launches of this size are host-bound), replayed."""
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
def lib():
    y = torch.addmm(b, x, w.t())
    parser_ff._act(y, y, rows, 1, H)
def lib512():
    y = x512 @ w512_kn
    parser_ff._act_bwd(y, act, y, rows, 1, H)
from vlgae_amd import align
wkn = (torch.randn(H, 800, generator=g) / 16).to(dev, bf)
out800 = torch.empty(rows, 800, device=dev, dtype=bf)
def lib_kn():
    torch.mm(x, wkn, out=out800)
outb = torch.empty(rows, H, device=dev, dtype=bf)
from vlgae_amd import encoders
rng = encoders.DeviceRng(3, dev)
def two_fwd():
    parser_ff._linear_act(x, w, b, out, rng=rng, p=0.3)
    parser_ff._linear_act(out, w, b, outb)
def chain_fwd():
    parser_ff._linear_act_chain2(x, parser_ff._stage(w, out, bias=b, rng=rng, p=0.3), parser_ff._stage(w, outb, bias=b))
def two_bwd():
    parser_ff._linear_act_bwd(x, w, act, out, rng=rng, p=0.3)
    parser_ff._linear_act_bwd(out, w, act, outb, J=4, total=tot, swap=True)
def chain_bwd():
    parser_ff._linear_act_chain2(x, parser_ff._stage(w, out, act=act, rng=rng, p=0.3), parser_ff._stage(w, outb, act=act, J=4, total=tot, swap=True), backward=True)
def lib32():
    torch.mm(x32, w32, out=out)
    parser_ff._act_bwd(out, act, out, rows, 1, H)
mb = rows * H * 2 * 2 / 1e6
for name, fn, mbytes in (("fused plain", lambda: parser_ff._linear_act(x, w, b, out), mb),
                         ("fused 2 blocks + residual", lambda: parser_ff._linear_act(x, w2, b2, out2, nb=2, residual=x, om=2, oy=1), mb * 1.5 + mb / 2),
                         ("fused backward J=1", lambda: parser_ff._linear_act_bwd(x, w, act, out), mb * 1.5),
                         ("fused backward J=4 swap sum", lambda: parser_ff._linear_act_bwd(x, w, act, out, J=4, total=tot, swap=True), mb * 1.5),
                         ("fused backward k=512", lambda: parser_ff._linear_act_bwd(x512, w512, act, out), mb * 2),
                         ("library k=512 + ff_act_bwd", lib512, mb * 2),
                         ("fused backward k=32", lambda: parser_ff._linear_act_bwd(x32, w32, act, out, w_kn=True), mb * 17 / 16),
                         ("library k=32 + ff_act_bwd", lib32, mb * 17 / 16),
                         ("linear_kn 256 -> 800 k=256", lambda: align.linear_kn(x, wkn, out=out800), rows * (256 + 800) * 2 / 1e6),
                         ("library 256 -> 800 k=256", lib_kn, rows * (256 + 800) * 2 / 1e6),
                         ("two forward launches (dropout | plain)", two_fwd, mb * 2),
                         ("chain2 forward", chain_fwd, mb * 1.5),
                         ("two backward launches (dropout | J=4 swap sum)", two_bwd, mb * 3),
                         ("chain2 backward", chain_bwd, mb * 2.5),
                         ("library GEMM + ff_act", lib, mb * 2)):
    if len(sys.argv) > 2 and sys.argv[2] not in name: continue
    us = t(fn)
    print("%-30s %7.1f us   %6.2f TB/s of its own bytes" % (name, us, mbytes / us))
