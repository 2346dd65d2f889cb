"""The grouped small products of the parser feed-forwards' passes alone (vlg_small_gemm_group), graph-timed: each group as launched in the training
step (B = 256, L = 40, H = 256, E = 800, h = 256, T = 45, r = 16, nb = 150), then every product of the group on its own -- which one sets the launch's time.
    python tools/time_small_group.py"""
import sys, torch
sys.path.insert(0, '.')
from vlgae_amd.align import SmallMatmulGroup
dev = torch.device('cuda:0')
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev, bf)
def t(fn, n=20, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): fn()
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / (n * reps)
B, L, H, E, h, T, r, nb = 256, 40, 256, 800, 256, 45, 16, 150
Ms = T + 3
# forward group: context term, token / root / decision rows, folded bottlenecks, folded projections
cmean, Wc, bh = R(B, h), R(H, h), R(H)
tok, Wtok = R(T, 16), R(H, 16)
W1s, W0s, b0s, b1s = R(4, H, nb), R(4, nb, H), R(4, nb), R(4, H)
PW, W2, b2, Pb = R(6 * r, H), R(H, H), R(H), R(6 * r)
ones4, ones1 = torch.ones(4, 1, device=dev, dtype=bf), torch.ones(1, device=dev, dtype=bf)
fwd = [("context term [B,h]x[h,H]", lambda G: G.add(cmean, Wc.t(), bias=bh)),
       ("token rows [T,16]x[16,H]", lambda G: G.add(tok, Wtok.t(), bias=bh)),
       ("folded bottlenecks 4x[H,nb]x[nb,H]", lambda G: G.add(W1s, W0s)),
       ("folded bottleneck biases", lambda G: G.add(W1s, b0s.unsqueeze(-1), rank1=(b1s, ones4))),
       ("folded projections [6r,H]x[H,H]", lambda G: G.add(PW, W2)),
       ("folded projection biases", lambda G: G.add(PW, b2.unsqueeze(-1), rank1=(Pb, ones1)))]
# backward group 1 (the small rows' cotangent), group 2 (weight space)
g_small, Wp, A5s = R(4 * Ms, 4 * r), R(6 * r, H), R(4 * Ms, H)
onesM = torch.ones(4 * Ms, 1, device=dev, dtype=bf)
bwd1 = [("small rows' input gradient [4Ms,4r]x[4r,H]", lambda G: G.add(g_small, Wp[2 * r:])),
        ("small rows' weight gradient [4r,4Ms]x[4Ms,H]", lambda G: G.add(g_small.t(), A5s)),
        ("small rows' bias gradient", lambda G: G.add(onesM.t(), g_small))]
dWp, dbp, gc, dWeff, dbeff = R(6 * r, H), R(6 * r), R(B, H), R(4, H, H), R(4, H)
gs = R(T, H)
bwd2 = [("dPW = dWp W2^T + dbp b2^T", lambda G: G.add(dWp, W2.t(), rank1=(dbp, b2))),
        ("linear2.w = PW^T dWp [H,6r]x[6r,H]", lambda G: G.add(PW.t(), dWp)),
        ("linear2.b", lambda G: G.add(PW.t(), dbp.unsqueeze(-1))),
        ("context weight gc^T cmean [H,B]x[B,h]", lambda G: G.add(gc.t(), cmean)),
        ("context input gc Wc [B,H]x[H,h]", lambda G: G.add(gc, Wc, alpha=1.0 / L)),
        ("token weight gs^T tok [H,T]x[T,16]", lambda G: G.add(gs.t(), tok)),
        ("token input gs W [T,H]x[H,16]", lambda G: G.add(gs, Wtok)),
        ("unfold dW1 = dWeff W0^T + dbeff b0^T  4x[H,H]x[H,nb]", lambda G: G.add(dWeff, W0s.transpose(1, 2), rank1=(dbeff, b0s))),
        ("unfold dW0 = W1^T dWeff  4x[nb,H]x[H,H]", lambda G: G.add(W1s.transpose(1, 2), dWeff)),
        ("unfold db0 = W1^T dbeff", lambda G: G.add(W1s.transpose(1, 2), dbeff.unsqueeze(-1)))]
for name, items in (("forward group", fwd), ("backward group 1", bwd1), ("backward group 2", bwd2)):
    def whole():
        G = SmallMatmulGroup()
        for _, f in items: f(G)
        G.launch()
    print("%-58s %6.1f us" % (name + " (%d products)" % len(items), t(whole)))
    for nm, f in items:
        def one():
            G = SmallMatmulGroup(); f(G); G.launch()
        print("    %-54s %6.1f us" % (nm, t(one)))
