"""Secondary single-GPU measurements appended to bench.py's JSON line (rank 0, N = 1 only).

None of them is the headline `value`; each entry says what it timed.  Kernel-only entries use HIP events on the launch
stream around back-to-back launches; "host API" entries include the Python / autograd overhead of the drop-in classes.
"""
import time

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0
MFMA_BF16_PEAK_TFLOPS = 2500.0
N_CU, SIMD_PER_CU, TRANS_LANES_PER_CLK, CLOCK_GHZ = 256, 4, 8, 2.4


def _events(fn, n, warm, dev):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e-3 / n      # seconds per call


def timed(fn, n, dev):
    """ms per call: median of three windows of n calls (a host hiccup inside one window -- an allocator refill,
    a scheduler tick -- otherwise lands in a host-bound entry as a 10x outlier)."""
    for _ in range(10):
        fn()
    wins = []
    for _ in range(3):
        wins.append(_events(fn, n, 0, dev) * 1e3)
    return sorted(wins)[1]


def dp_raw(B, L, dtype, dev, seed=1000):
    """Raw C-ABI launch closure of the fused DMV1o inside+outside kernel on fresh synthetic potentials."""
    import bench
    from vlgae_amd import _C
    import vlgae_amd.torch_struct as ts
    lib = _C.lib()
    N = L + 1
    dec, attach, root = bench.synth(B, L, seed, dev, torch.float32)
    md32, ma32 = ts.DMV1o.merge(dec, attach, root)
    md, ma = md32.to(dtype).contiguous(), ma32.to(dtype).contiguous()
    lengths = torch.full((B,), L, dtype=torch.long, device=dev)
    logZ = torch.empty(B, dtype=torch.float32, device=dev)
    gdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dev)
    gatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dev)
    ws_bytes = lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, B, N, 0)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    code = _C.BF16 if dtype == torch.bfloat16 else _C.F32
    sp = _C.ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    p = [_C.ptr(x) for x in (md, ma, lengths, logZ, gdec, gatt, ws)]
    keep = (md, ma, lengths, logZ, gdec, gatt, ws)

    def launch():
        rc = lib.vlg_dmv1o_inside_outside(p[0], p[1], p[2], B, N, code, 0, None, p[3], p[4], p[5], p[6], ws_bytes, sp)
        if rc:
            _C.check(rc, "dmv1o_inside_outside")
    launch.keep = keep
    launch.ws_bytes = ws_bytes
    return launch


def dp_entry(B, L, dtype_name, dev, n=100):
    import bench
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    launch = dp_raw(B, L, dtype, dev)
    sec = _events(launch, n, 10, dev)
    logZ, gatt = launch.keep[3], launch.keep[5]
    assert bool(torch.isfinite(logZ).all())
    assert abs(float(gatt.sum().item()) - B * L) < 1e-3 * B * L
    N = L + 1
    alg = bench.algorithmic_bytes(B, N, 2 if dtype_name == "bf16" else 4)
    ops = bench.exp_class_ops(np.full(B, L))
    exp_peak = N_CU * SIMD_PER_CU * TRANS_LANES_PER_CLK * CLOCK_GHZ * 1e9
    # counter traffic of THIS workload from the committed PMC passes (tools/prof_headline.sh -> tools/dp_workload.py), keyed by workload and
    # labelled with the kernel-source hash it was taken at; None only when no committed profile holds the workload
    traffic, traffic_src = bench.pmc_traffic(f"dmv1o_B{B}_L{L}_{dtype_name}")
    return {"us": sec * 1e6, "sentences_per_s": B / sec,
            "workload": f"DMV1o inside+outside (Log), B={B} L={L} N={N}, potentials stored {dtype_name}, raw C-ABI launches",
            "workspace_bytes": int(launch.ws_bytes),
            "roofline": {"bound": "hbm", "achieved": alg / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / sec / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": alg, "traffic": traffic,
                         "traffic_over_algorithmic": None if traffic is None else traffic / alg, "traffic_source": traffic_src},
            "exp_rate": {"achieved_Gops": ops / sec / 1e9, "peak_Gops": exp_peak / 1e9, "frac": ops / sec / exp_peak}}


def dp_capacity_entry(B, L, dtype_name, dev):
    """What bounds the headline: at B = 256 there is ONE workgroup per CU and a launch takes the critical path of one sentence.  Reported beside it
    (not as `value`): single launches of 2 B and 8 B sentences (8 B = configs[2]'s batch on one GPU), where a CU holds a second workgroup --
    the chip's capacity for this DP.  (Two B-sentence launches alternating on two HIP streams do NOT get there: 152 us per batch, measured --
    the queues do not interleave their workgroups the way one launch does.)"""
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    res = {}
    for mult in (2, 8):
        launch = dp_raw(mult * B, L, dtype, dev, seed=2000 + mult)
        sec = _events(launch, 50, 10, dev)
        res[f"one_launch_of_{mult * B}"] = {"us": sec * 1e6, "sentences_per_s": mult * B / sec}
        del launch
    res["note"] = ("the headline `value` is ONE B-sentence launch at a time, as BASELINE.json quotes it (B = 256: one workgroup per CU, latency-bound); "
                   "these figures show what a second workgroup per CU adds")
    return res


_KERNEL_EVIDENCE = None


def kernel_evidence():
    """The newest committed per-kernel profile (profiles/r*_kernels.json, written by tools/prof_kernels.sh + pmc_summary.py)."""
    global _KERNEL_EVIDENCE
    if _KERNEL_EVIDENCE is None:
        import glob
        import json
        import os
        root = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "profiles")   # <repo>/profiles
        files = sorted(glob.glob(os.path.join(root, "r*_kernels.json")))
        _KERNEL_EVIDENCE = (None, {}) if not files else (os.path.basename(files[-1]), json.load(open(files[-1])))
    return _KERNEL_EVIDENCE


def mfma_busy(kernel):
    """Matrix-core evidence for the kernel a bench entry LAUNCHES, by that kernel's own name (the prefix before the template
    arguments must match a profiled kernel exactly): {"frac": SQ_VALU_MFMA_BUSY_CYCLES / SIMD-cycles, ...} from the newest committed
    profile -- counters cannot be collected inside this run (rocprofv3 --pmc needs its own passes).  A name the profile does not
    hold gives {"frac": None, "reason": ...}: no figure of a replaced kernel is ever quoted."""
    name, d = kernel_evidence()
    if name is None:
        return {"frac": None, "reason": "no profiles/r*_kernels.json committed"}
    hits = {k: e for k, e in d.get("kernels", {}).items() if k.split("<")[0] == kernel and "mfma_busy" in e}
    if not hits:
        return {"frac": None, "reason": f"profiles/{name} holds no kernel named {kernel}: the entry's kernel was not profiled"}
    k, e = max(hits.items(), key=lambda kv: kv[1].get("avg_us", 0.0) * kv[1].get("calls", 1))
    out = {"frac": e["mfma_busy"], "kernel": k, "avg_us_under_profiler": e.get("avg_us"), "source": f"profiles/{name} (kernel source {d.get('kernel_source_id')})"}
    if "hbm_bytes_per_launch" in e:
        out["hbm_bytes_per_launch"] = e["hbm_bytes_per_launch"]
    if "wait_any_frac" in e:
        out["wait_any_frac"] = e["wait_any_frac"]
    return out


def run_all(out, args, h, dev):
    with torch.autograd.set_multithreading_enabled(False):   # see the api_path entry
        _run_all(out, args, h, dev)
    out["autograd_mode"] = "backward on the calling thread (torch.autograd.set_multithreading_enabled(False)); one process per GPU"


def _run_all(out, args, h, dev):
    import vlgae_amd.torch_struct as ts
    from vlgae_amd.torch_struct import functional as Fn
    from vlgae_amd import align
    B, L, N = h.B, h.L, h.N
    md, ma, lengths = h.md, h.ma, h.lengths
    in_dtype = h.in_dtype

    # ---- the same headline step with fp32-stored potentials (what DMV1o.merge emits, distributions.py:253-265) ----
    other = "f32" if args.dtype == "bf16" else "bf16"
    out["headline_" + other] = dp_entry(B, L, other, dev)

    try:   # the chip's capacity for this DP beside the latency-bound headline
        out["dp_capacity"] = dp_capacity_entry(B, L, args.dtype, dev)
    except Exception as e:
        out["dp_capacity"] = {"error": repr(e)[:200]}

    # ---- configs[3]: B=256 L=80 long-sentence stress, fused inside+outside ----
    if L != 80:
        out["long_sentence"] = dp_entry(B, 80, args.dtype, dev, n=30)

    # ---- the same step through the drop-in Python API (DMV1o(...).partition + autograd.grad) ----
    # torch runs a GPU graph's backward on a per-device engine thread; the hand-off (a condition-variable wake-up) costs
    # 50-120 us per call on these hosts and varies box to box, which is more than this kernel takes.  One process per GPU
    # has no use for that thread: torch.autograd.set_multithreading_enabled(False) keeps backward on the calling thread
    # (INTEGRATION.md section 2).  Every autograd-timed entry below runs that way; this entry reports both.
    d_, a_ = md.detach().requires_grad_(), ma.detach().requires_grad_()

    def api_rate(n_api=500):
        for _ in range(100):
            torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n_api):
            torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
        torch.cuda.synchronize(dev)
        return B * n_api / (time.perf_counter() - t0)
    with torch.autograd.set_multithreading_enabled(True):
        engine = api_rate()
    out["api_path"] = {"sentences_per_s": api_rate(), "sentences_per_s_engine_thread": engine,
                       "what": "DMV1o([dec,attach],lengths).partition.sum() + torch.autograd.grad, 1 GPU; backward on the calling "
                               "thread (torch.autograd.set_multithreading_enabled(False)) | on torch's default engine thread"}

    # ---- Viterbi decode of the same batch: Max-semiring inside + back-pointer walk -> head vector (joint.py:256-258) ----
    sec = _events(lambda: Fn.dmv1o_decode(md, ma, lengths), 50, 5, dev)
    out["decode"] = {"us": sec * 1e6, "sentences_per_s": B / sec, "what": "dmv1o_decode: best tree as heads [B,N], on device"}
    sec = _events(lambda: ts.DMV1o([md, ma], lengths).marginals_and_heads(), 50, 5, dev)
    out["marginals_and_heads"] = {"us": sec * 1e6,
                                  "what": "arc marginals + Viterbi heads of one batch (joint.py:251-258): ONE launch, grid (B, 2) "
                                          "(vlg_dmv1o_marginals_viterbi; two HIP streams before round 4's second half and beyond N = 44)"}
    sec = _events(lambda: Fn.dmv1o_viterbi(md, ma, lengths), 50, 5, dev)
    out["viterbi_with_counts"] = {"us": sec * 1e6, "what": "Max semiring: best score + tree counts (d max / d potentials) + heads, one launch "
                                                           "(inside pass + back-pointer walk; the one-hot outside replay is no longer run)"}

    # ---- the region x word alignment that feeds / consumes the DP (joint.py:406-419) ----
    # features as SURVEY 8(d) specifies: 2048-d region / 768-d word vectors through fixed-seed Linear(-> 128), no bias
    Q, V, d = 2 * N, args.regions, 128
    g = torch.Generator().manual_seed(5)
    w_vis = (torch.randn(2048, d, generator=g) / 2048 ** 0.5).to(dev)
    w_txt = (torch.randn(768, d, generator=g) / 768 ** 0.5).to(dev)
    vis = (torch.randn(B, V, 2048, generator=g).to(dev) @ w_vis).to(in_dtype).contiguous()
    txt = (torch.randn(B, Q, 768, generator=g).to(dev) @ w_txt).to(in_dtype).contiguous()
    del w_vis, w_txt
    for full in (True, False):
        kw = dict(full=full, max_v=not full, max_q=not full)
        sec = _events(lambda: align.bilinear_align(txt, vis, **kw), 100, 20, dev)
        flops = 2.0 * B * B * Q * V * d
        esz = 2 if in_dtype == torch.bfloat16 else 4
        byts = (B * Q + B * V) * d * esz + (B * B * Q * V * 4 if full else (B * B * (Q + V)) * 4)
        out["align_full" if full else "align_fused_max"] = {
            "sentences_per_s": B / sec, "ms": sec * 1e3, "TFLOP/s": flops / sec / 1e12,
            "frac_mfma_bf16_peak": flops / sec / 1e12 / MFMA_BF16_PEAK_TFLOPS if in_dtype == torch.bfloat16 else None,
            "GB/s": byts / sec / 1e9, "frac_hbm": byts / sec / 1e9 / HBM_PEAK_GBS,
            "mfma_busy": mfma_busy("align_full_kernel" if full else "align_max_kernel") if in_dtype == torch.bfloat16 else None,
            "shape": f"B=A={B} Q={Q} V={V} d={d} {args.dtype} in (2048-d / 768-d features through fixed-seed Linear->128), fp32 out"}

    # ---- the plain projections around the contraction (`vis_mlp_pre_matching` joint.py:136-138,175; the visual
    #      encoder's box MLP box_rel.py:29-40; word / child / parent encoders joint.py:218-221): nn.Linear shapes, LEFT to the
    #      library (hipBLASLt through torch) by design -- measured here so that the choice is on record ----
    def gemm_entry(m, k, n):
        a_ = torch.randn(m, k, generator=g).to(dev, in_dtype)
        w_ = torch.randn(k, n, generator=g).to(dev, in_dtype)
        sec_ = _events(lambda: a_ @ w_, 50, 10, dev)
        fl = 2.0 * m * k * n
        return {"ms": sec_ * 1e3, "TFLOP/s": fl / sec_ / 1e12,
                "frac_mfma_bf16_peak": fl / sec_ / 1e12 / MFMA_BF16_PEAK_TFLOPS if in_dtype == torch.bfloat16 else None,
                "GB/s": (m * k + k * n + m * n) * (2 if in_dtype == torch.bfloat16 else 4) / sec_ / 1e9, "shape": f"[{m},{k}] x [{k},{n}] {args.dtype}"}
    out["library_gemms"] = {
        "vis_box_mlp": gemm_entry(B * V, 2048, 256),          # region features -> hidden (box_rel.py:29-40, n_in 2048, n_hidden 256)
        "vis_mlp_pre_matching": gemm_entry(B * V, 256, 128),  # hidden -> matching space (joint.py:136-138)
        "word_child_parent": gemm_entry(B * N, 256, 128),     # word encodings -> child / parent (joint.py:218-221)
        "note": "library GEMMs (torch -> hipBLASLt), not part of this package: small-N projections are memory-bound on the "
                "activation read; the frac_mfma figure is therefore low by construction"}

    # ---- backward of the materialised tensor (what autograd runs when the reference's own loss consumes gather_logit_simple's
    #      [B,A,Q,V] output): both feature gradients, each kernel reads the 4 B * B*A*Q*V cotangent once ----
    cot = torch.randn(B, B, Q, V, generator=torch.Generator(device=dev).manual_seed(7), device=dev)
    # the MASKED call is the one the model makes (root slots of both halves of txt masked, joint.py:204); the unmasked one beside it
    tm_b = torch.ones(B, Q, dtype=torch.bool, device=dev)
    tm_b[:, 0] = False
    tm_b[:, Q // 2] = False
    vm_b = torch.ones(B, V, dtype=torch.bool, device=dev)
    sec_unmasked = _events(lambda: align.bilinear_align_backward(cot, txt, vis), 20, 3, dev)
    sec = _events(lambda: align.bilinear_align_backward(cot, txt, vis, tm_b, vm_b), 20, 3, dev)
    byts = 2.0 * cot.numel() * 4
    out["align_backward"] = {"mfma_busy": ({"align_bwd_split_kernel": mfma_busy("align_bwd_split_kernel"), "align_bwd_split2_kernel": mfma_busy("align_bwd_split2_kernel")}
                                           if in_dtype == torch.bfloat16 else None),
                             "ms": sec * 1e3, "unmasked_ms": sec_unmasked * 1e3, "GB/s": byts / sec / 1e9, "frac_hbm": byts / sec / 1e9 / HBM_PEAK_GBS,
                             "TFLOP/s": 2 * 2.0 * B * B * Q * V * d / sec / 1e12,
                             "shape": f"B=A={B} Q={Q} V={V} d={d} {args.dtype} features, fp32 cotangent [B,A,Q,V] "
                                      f"({cot.numel() * 4 / 1e6:.0f} MB, read twice: once per gradient)"}
    del cot

    # the two consumers on the training path: attention-fuse (joint.py:670-674) and the grounding loss on the
    # fused maxima (joint.py:439-491), forward + gradients, through the host API
    hdim = 256
    mk = lambda *shape: torch.randn(*shape, generator=g).to(dev, in_dtype).requires_grad_(True)
    f_vis, f_txt, f_mid, f_enc = mk(B, V, d), mk(B, N, d), mk(B, V, hdim), mk(B, L, hdim)
    ln_w, ln_b = torch.ones(hdim, device=dev, requires_grad=True), torch.zeros(hdim, device=dev, requires_grad=True)
    dout = torch.randn(B, L, hdim, generator=g).to(dev)
    leaves = [f_vis, f_txt, f_mid, f_enc, ln_w, ln_b]
    out["attention_fuse"] = {
        "fwd_ms": timed(lambda: align.attention_fuse(f_vis.detach(), f_txt.detach(), f_mid.detach(), f_enc.detach(),
                                                     ln_w.detach(), ln_b.detach(), 1e-5), 50, dev),
        "fwd_bwd_ms": timed(lambda: torch.autograd.grad(align.attention_fuse(*leaves, 1e-5), leaves, dout), 50, dev),
        "mfma_busy": {"fwd": mfma_busy("attn_fuse_mfma_kernel"), "bwd_words": mfma_busy("attn_fuse_bwd_words_kernel")},
        "shape": f"B={B} L={L} V={V} d={d} h={hdim} {args.dtype} in; host API incl. autograd overhead"}
    tmask = torch.ones(B, Q, dtype=torch.bool, device=dev)
    tmask[:, 0] = tmask[:, N] = False
    vmask = torch.ones(B, V, dtype=torch.bool, device=dev)
    marg = torch.rand(B, Q, generator=g).to(dev) * tmask
    g_txt, g_vis = mk(B, Q, d), mk(B, V, d)

    def ground():
        total, _ = align.grounding_loss_factor_ce(g_txt, g_vis, tmask, vmask, marg, B * L, 1.0)
        return torch.autograd.grad(total, [g_txt, g_vis])
    out["grounding_loss"] = {"fwd_bwd_ms": timed(ground, 10, dev),
                             "mfma_busy": ({"align_argmax_kernel": mfma_busy("align_argmax_kernel"), "ground_bwd_ws_kernel": mfma_busy("ground_bwd_ws_kernel")}
                                           if in_dtype == torch.bfloat16 else None),
                             "shape": f"B=A={B} Q={Q} V={V} d={d} {args.dtype} in; loss + gradients, no [B,A,Q,V] tensor"}
    out["grounding_decode"] = {
        "ms": timed(lambda: align.grounding_decode(g_txt.detach(), g_vis.detach(), tmask, vmask), 20, dev),
        "shape": f"B=A={B} Q={Q} V={V} d={d} {args.dtype} in; alignment (diag + max_v) + top-5 / image arg-max, joint.py:512-596"}

    # the same two at the reference's shipped factor layout (config/data/vlparse.yaml: 36 boxes -> obj 36 + rel 1296 +
    # attr 36 + img 1 = 1369 columns, batch 64): 29 region groups per image, fewer captions than CUs
    Bs, Vs = 64, 1369
    s_txt, s_vis = mk(Bs, Q, d), mk(Bs, Vs, d)
    s_tmask, s_vmask = tmask[:Bs], torch.ones(Bs, Vs, dtype=torch.bool, device=dev)
    s_marg = marg[:Bs]

    def ground_shipped():
        total, _ = align.grounding_loss_factor_ce(s_txt, s_vis, s_tmask, s_vmask, s_marg, Bs * L, 1.0)
        return torch.autograd.grad(total, [s_txt, s_vis])
    out["shipped_layout"] = {
        "grounding_loss_fwd_bwd_ms": timed(ground_shipped, 10, dev),
        "grounding_decode_ms": timed(lambda: align.grounding_decode(s_txt.detach(), s_vis.detach(), s_tmask, s_vmask), 10, dev),
        "shape": f"B=A={Bs} Q={Q} V={Vs} d={d} {args.dtype} in"}
    del s_txt, s_vis
    # the attention fuse at that layout (1369 keys per image: the key-split kernels, csrc/vlg_attn.hip), forward and forward + adjoint,
    # as captured HIP graphs (device time: a handful of launches of 5-45 us each is host-bound when enqueued eagerly)
    try:
        w_vis, w_txt, w_mid, w_enc = mk(Bs, Vs, d), mk(Bs, N, d), mk(Bs, Vs, hdim), mk(Bs, L, hdim)
        w_leaves = [w_vis, w_txt, w_mid, w_enc, ln_w, ln_b]
        w_dout = dout[:Bs].contiguous()

        def graph_ms(fn, n=30):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    fn()
            torch.cuda.current_stream(dev).wait_stream(side)
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                keep = fn()
            for _ in range(5):
                gr.replay()
            sec = _events(gr.replay, n, 3, dev)
            del keep, gr
            return sec * 1e3
        with torch.autograd.set_multithreading_enabled(False):
            fwd = graph_ms(lambda: align.attention_fuse(*(x.detach() for x in w_leaves), 1e-5))
            fwd_bwd = graph_ms(lambda: torch.autograd.grad(align.attention_fuse(*w_leaves, 1e-5), w_leaves, w_dout))
        flops_f = 2.0 * Bs * L * Vs * (d + hdim)
        out["shipped_layout"]["attention_fuse"] = {
            "fwd_ms": fwd, "fwd_bwd_ms": fwd_bwd, "fwd_TFLOP/s": flops_f / (fwd * 1e-3) / 1e12,
            "key_bytes_GB/s": Bs * Vs * (d + hdim) * (2 if in_dtype == torch.bfloat16 else 4) / (fwd * 1e-3) / 1e9,
            "shape": f"B={Bs} L={L} V={Vs} d={d} h={hdim} {args.dtype} in; HIP-graph replays (forward: split + merge + combine launches; "
                     "adjoint: combine + sweep + d_txt + regions + affine, the forward's merged records handed over)"}
        del w_vis, w_txt, w_mid, w_enc
    except Exception as e:   # never costs the line
        out["shipped_layout"]["attention_fuse"] = {"error": repr(e)[:200]}

    # arc encoder's trilinear term (joint.py:282-284): M = B * (L + 1) rows, 128^3 weights
    a_child, a_parent = mk(B, N, d), mk(B, N, d)
    a_w1 = (torch.randn(d, d, d, generator=g) / d).to(dev, in_dtype).requires_grad_(True)
    a_dout = torch.randn(B, N, d, generator=g).to(dev)
    out["arc_trilinear"] = {
        "fwd_ms": timed(lambda: align.arc_trilinear(a_child.detach(), a_w1.detach(), a_parent.detach()), 20, dev),
        "fwd_bwd_ms": timed(lambda: torch.autograd.grad(align.arc_trilinear(a_child, a_w1, a_parent),
                                                        [a_child, a_w1, a_parent], a_dout), 10, dev),
        "mfma_busy": {"tri2_kernel": mfma_busy("tri2_kernel"), "tri_dw2_kernel": mfma_busy("tri_dw2_kernel")},
        "shape": f"M={B * N} X=H=Y={d} {args.dtype} in; einsum('bcx,xhy,bcy->bch') without the [M,H,Y] intermediate"}


    # ---- round 3: encoder weight gradients (split-K), the language-side feature stage, the fused score construction ----
    try:
        out.update(round3_entries(B, L, in_dtype, dev, g))
    except Exception as e:
        out["round3_entries_error"] = repr(e)[:300]

    # ---- configs[4]: the chained training-step hot path (vlgae_amd/train_step.py), eager and as one captured HIP graph ----
    try:
        out["train_step"] = train_step_entry(B, L, V, in_dtype, dev)
    except Exception as e:   # a capture failure must not cost the headline line
        out["train_step"] = {"error": repr(e)[:300]}


def round3_entries(B, L, dtype, dev, g):
    import vlgae_amd.torch_struct as ts
    from vlgae_amd import align, langfeat, scorer
    res = {}
    N, d, hdim, T, r = L + 1, 128, 256, 45, 16
    bf = torch.bfloat16
    rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    # weight / bias gradient of the three encoders' fused Linear: dY [M,384]^T x [M,256] over M = B*N rows
    M = B * N
    dy, x = rnd(M, 3 * d).to(bf), rnd(M, hdim).to(bf)
    res["linear_wgrad"] = {
        "mfma_busy": mfma_busy("gemm_tn_kernel"),
        "ms": timed(lambda: align.linear_wgrad(dy, x), 50, dev), "library_ms": timed(lambda: (dy.t() @ x, dy.float().sum(0)), 50, dev),
        "shape": f"dY [{M},{3 * d}]^T x [{M},{hdim}] bf16 -> fp32 [{3 * d},{hdim}] + bias [{3 * d}]; split-K over the rows on "
                 "ds_read_b64_tr_b16 + bf16 MFMA, fixed-order partial sums; library = torch matmul (4 output tiles on 256 CUs) + sum"}
    # lang_feat_max_tree (joint.py:235-292), forward + gradients, both DPs included
    xw = rnd(B, L, hdim, sc=0.5).to(dtype).requires_grad_(True)
    xw32 = xw.detach().float().requires_grad_(True)
    params32 = None
    params = [rnd(3 * d, hdim, sc=hdim ** -0.5).to(bf).requires_grad_(True), rnd(3 * d, sc=0.1).to(bf).requires_grad_(True),
              rnd(d, d, d, sc=1.0 / d).to(bf).requires_grad_(True), rnd(d, d, sc=d ** -0.5).to(bf).requires_grad_(True),
              rnd(d, sc=0.1).to(bf).requires_grad_(True)]
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1).to(dev)
    attach, root = torch.randn(B, L, L, 2, generator=g).to(dev), torch.randn(B, L, generator=g).log_softmax(-1).to(dev)
    md, ma = ts.DMV1o.merge(dec, attach, root)
    md, ma = md.to(dtype), ma.to(dtype)
    lengths = torch.full((B,), L, dtype=torch.long, device=dev)
    dout = rnd(B, 2 * N, d).to(bf)

    def lf():
        txt, _, _ = langfeat.lang_feat_max_tree(xw, lengths, md, ma, *params)
        return torch.autograd.grad(txt, [xw] + params, dout)
    def lf32():   # float32 features and parameters: computed in float32 end to end (the reference's `precision: 32`)
        txt, _, _ = langfeat.lang_feat_max_tree(xw32, lengths, md, ma, *params32)
        return torch.autograd.grad(txt, [xw32] + params32, dout.float())
    params32 = [p_.detach().float().requires_grad_(True) for p_ in params]
    res["lang_feat_max_tree"] = {"fwd_bwd_ms": timed(lf, 20, dev), "fwd_bwd_ms_float32_features": timed(lf32, 20, dev),
                                 "shape": f"B={B} L={L} h={hdim} d={d}: DMV marginals || Viterbi heads, masked-mean root row, word|child|parent "
                                          "encoders as one GEMM, arc encoder, txt [B,2N,d] bf16; gradients to x and every parameter (joint.py:235-292)"}
    # score construction feeding the DP (ldndmv.py:184-209): fused vs the reference's formulation in torch ops, both followed by the same DP
    ins = [rnd(B, L, 2, 2, r, sc=0.5).requires_grad_(True), rnd(T, 2, 2, r, sc=0.5).requires_grad_(True), rnd(B, L, 2, 2, r, sc=0.5).requires_grad_(True),
           rnd(2, 2, 2, r, sc=0.5).requires_grad_(True), torch.randn(T, generator=g).log_softmax(-1).to(dev).requires_grad_(True)]
    token = torch.randint(0, T, (B, L), generator=g).to(dev)

    def fused():
        smd, sma = scorer.ndmv_potentials(*ins, token)
        return torch.autograd.grad(ts.DMV1o([smd, sma], lengths).partition.sum(), ins)

    def torch_glue():
        x1, x2, y1, y2, root_rule = ins
        attach_rule = torch.einsum("bhdve,cdve->bhcdv", x1, x2).log_softmax(2)
        ap = attach_rule.gather(2, token.reshape(B, 1, L, 1, 1).expand(B, L, L, 2, 2))
        left, right = torch.tril(torch.ones(L, L, device=dev), diagonal=-1), torch.triu(torch.ones(L, L, device=dev), diagonal=1)
        ap = ap[..., 0, :] * left.unsqueeze(0).unsqueeze(-1) + ap[..., 1, :] * right.unsqueeze(0).unsqueeze(-1)
        dc = torch.einsum("bhdve,kdve->bhkdv", y1, y2).permute(0, 1, 3, 4, 2).log_softmax(-1)
        rt = torch.gather(root_rule.unsqueeze(0).expand(B, -1), 1, token)
        m1, m2 = ts.DMV1o.merge(dc, ap, rt)
        return torch.autograd.grad(ts.DMV1o([m1, m2], lengths).partition.sum(), ins)
    res["score_construction"] = {
        "fused_fwd_bwd_ms": timed(fused, 30, dev), "torch_ops_fwd_bwd_ms": timed(torch_glue, 30, dev),
        "shape": f"B={B} L={L} T={T} r={r} fp32: scorers' projected inputs -> merged potentials -> DMV1o partition -> gradients back "
                 "to the inputs (ldndmv.py:184-209 + dmv.py:19-66); fused = vlg_ndmv_potentials (no [B,L,T,2,2] rule table), "
                 "torch_ops = the reference's einsum / log_softmax / gather / tril-triu / merge lines; both use this package's DP kernel"}
    return res


def train_step_entry(B, L, V, dtype, dev, wiring="reference", factors=()):
    """configs[4]: vlgae_amd.train_step.build -- the function tests/test_gpu_parity.py::test_training_step_reference_wiring pins on
    fixtures made by the reference's own methods, from the frozen features to every trainable gradient -- eager and as one captured
    HIP graph.  factors: the visual factors beside the objects (the shipped model: rel, attr, img = 1369 columns at 36 regions).
    wiring="r3": round 3's chain (bench continuity only; it is not what the reference wires)."""
    from vlgae_amd import train_step
    step = train_step.build(B, L, V, dev, dtype=dtype, wiring=wiring, factors=factors)
    for _ in range(5):
        step()

    def wall(fn, n):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n * 1e3, (t1 - t0) / n * 1e3
    eager_ms, enqueue_ms = wall(step, 30)
    if wiring == "reference":
        what = ("one training step as the reference wires it (base.py:215-241, joint.py:658-711, ldndmv.py:171-216,260-285, fn.py:50-56), from "
                "the FROZEN features: VisBoxRelSimpleEncoder (box_fc" + ("".join(f" | {f}_fc" for f in factors if f != "img")) + " on [box ; mean box], 2048-d "
                "region features) + MLPEncoder (nn.Dropout p=0.33 drawn per step + Linear 800->256) -> vis_mlp_pre_matching -> lang_feat_word_only -> attention fuse -> [fused x] context mean + the parser's feed-forwards "
                "(vlgae_amd.parser_ff: head_ff / mid_ff / scorer projections, E=800 H=256 n_bottleneck=150 r=16, their dropout 0.33 / 0.3 drawn per step) -> score construction -> "
                "[un-fused x] lang_feat_max_tree (DMV1o marginals || one Viterbi pass, word|child|parent encoders with SharedDropout p=0.33 "
                "drawn per step, arc encoder) -> alignment maxima with the POS prior + grounding cross-entropies (ragged vis_mask) -> "
                "-DMV1o.max -> 0.5 mt + 0.5 dep -> / num_token -> gradients to every input feature and parameter")
    else:
        what = ("ROUND 3's chain (not the reference's wiring: fused x into lang_feat_max_tree, scorer inputs / vis_feat / fuse word "
                "features as leaves, no prior / dropout / alpha): score construction -> attention_fuse -> lang_feat_max_tree -> "
                "grounding loss -> -DMV1o.max -> gradients")
    res = {"eager_ms": eager_ms, "host_enqueue_ms": enqueue_ms, "wiring": wiring,
           "what": what + f" (vlgae_amd/train_step.py), B={B} L={L} R={V} regions, {getattr(step, 'shape', {}).get('V', V)} factor columns, d=128 h=256, "
                          "synthetic frozen features: random 800-d embeddings / 2048-d region features (BERT / Faster-RCNN weights are not in the container)",
           "sentences_per_s_eager": B / (eager_ms * 1e-3)}
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    with torch.cuda.graph(gr):
        step()
    for _ in range(5):
        gr.replay()
    graph_ms, _ = wall(gr.replay, 50)
    res.update(graph_ms=graph_ms, sentences_per_s_graph=B / (graph_ms * 1e-3),
               note="graph replay has no host work between kernels: graph_ms is the device time of the chain; "
                    "eager_ms - graph_ms is what the Python / autograd host path still costs")
    if wiring == "reference" and not factors and dtype != torch.float32:
        del gr, step
        try:   # the same step in float32 (the reference's `precision: 32`, config/trainer/train.yaml:20): split-K weight gradients on three bf16 products,
            # arc trilinear and arg-max alignment on two fp16 parts per operand (three MFMAs per product; float32-level results)
            c = train_step_entry(B, L, V, torch.float32, dev)
            res["float32"] = {"graph_ms": c["graph_ms"], "eager_ms": c["eager_ms"], "ratio_to_this_dtype": c["graph_ms"] / graph_ms}
        except Exception as e:
            res["float32"] = {"error": repr(e)[:200]}
        try:   # the same step at the shipped factor layout (add_rel / add_attr / add_image, B = 64 as config/data/vlparse.yaml:24-27 batches it)
            c = train_step_entry(64, L, V, dtype, dev, factors=("rel", "attr", "img"))
            res["shipped_factor_layout"] = {k: c[k] for k in ("graph_ms", "eager_ms", "sentences_per_s_graph", "what")}
        except Exception as e:
            res["shipped_factor_layout"] = {"error": repr(e)[:200]}
        try:   # the parser's feed-forwards alone (forward + backward), for the breakdown
            res["parser_feed_forward"] = parser_ff_ms(B, L, dtype, dev)
        except Exception as e:
            res["parser_feed_forward"] = {"error": repr(e)[:200]}
        try:
            c = train_step_entry(B, L, V, dtype, dev, wiring="r3")
            res["round3_chain"] = {"graph_ms": c["graph_ms"], "eager_ms": c["eager_ms"], "what": c["what"]}
        except Exception as e:
            res["round3_chain"] = {"error": repr(e)[:200]}
    return res


def parser_ff_ms(B, L, dtype, dev, E=800, h=256, Et=32, T=45, H=256, nb=150, r=16):
    """The parser's feed-forwards (ldndmv.py:174-205), forward + backward: vlgae_amd.parser_ff against the module-by-module torch
    formulation the reference's modules amount to."""
    from vlgae_amd import train_step
    from vlgae_amd import parser_ff
    g = torch.Generator().manual_seed(3)
    P = train_step.init_feed_forward(g, dev, dtype, E, h, Et, T, H, nb, r)
    emb = torch.randn(B, L, E, generator=g).to(dev, dtype).requires_grad_(True)
    x = torch.randn(B, L, h, generator=g).to(dev, dtype).requires_grad_(True)
    leaves = [emb, x] + list(P.values())

    def run(fn):
        outs = fn(P, emb, x)
        torch.autograd.grad([o.float().sum() for o in outs], leaves, allow_unused=True)
    return {"fused_ms": timed(lambda: run(parser_ff.parser_feed_forward), 20, dev),
            "module_by_module_torch_ms": timed(lambda: run(train_step.scorer_feed_forward), 20, dev),
            "what": f"emb [B,L,{E}] + fused encodings -> head_ff / mid_ff (H={H}, n_bottleneck={nb}) / scorer projections (r={r}) -> gradients; "
                    "eager, event-timed (the module-by-module form is host-bound: ~190 launches)"}
