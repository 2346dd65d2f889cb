#!/usr/bin/env python3
"""bench.py -- sentences/sec of the batched inside-outside hot path on MI355X.

    python bench.py                          # 1 GPU, B=256 L=40 (BASELINE.json configs[1])
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic root-merged potentials already resident
in HBM: the fused DMV1o inside+outside launch (Log semiring) producing logZ [B] and the expected counts
grad_dec [B,N,2,2,2] / grad_attach [B,N,N,2] -- what `torch.autograd.grad(DMV1o(...).partition.sum(), ...)`
costs in the reference (src/model/joint.py:254-255).  With N > 1 ranks every rank owns 256 sentences (weak
scaling, global batch 256*N, configs[2]) and each step ends with ONE RCCL all-reduce of the
marginal-loss gradient (batch-summed expected counts, optionally padded to a model-sized buffer with
--grad-mb), issued asynchronously so it overlaps the next step's kernel; all of them complete inside the
timed region.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
N_CU, SIMD_PER_CU = 256, 4
TRANS_LANES_PER_CLK = 8        # v_exp_f32 / v_log_f32: 8 cycles per wave64 instruction per SIMD -> 8 lanes/clk
CLOCK_GHZ = 2.4


def synth(B, L, seed, device, dtype):
    """Potentials shaped like the scorer's output (normalised log-probs), SURVEY.md section 8d."""
    g = torch.Generator().manual_seed(seed)
    dec = torch.randn(B, L, 2, 2, 2, generator=g).log_softmax(-1)
    attach = torch.randn(B, L, L, 2, generator=g)
    root = torch.randn(B, L, generator=g).log_softmax(-1)
    return dec.to(device=device, dtype=dtype), attach.to(device=device, dtype=dtype), root.to(device=device, dtype=dtype)


def algorithmic_bytes(B, N, in_bytes):
    """read dec+attach once, write logZ + both gradients once."""
    per = (2 * N * N + 8 * N) * in_bytes + 4 + (2 * N * N + 8 * N) * 4
    return per * B


def exp_class_ops(lengths):
    """inside N^3-N lse terms + ~2x that outside, per sentence with N = len+1."""
    n = lengths.astype(np.float64) + 1
    return float((3 * (n ** 3 - n)).sum())


def cpu_baseline(B, L, seed, budget_s):
    """The CPU oracle (C restatement of the reference algorithm, fp32, OpenMP over sentences) timed on this
    box's host cores on a bounded sample of the same workload."""
    import oracle
    oracle.build()
    threads = oracle.max_threads()
    B = max(B, 16 * threads)          # enough sentences per core for the OpenMP loop to scale
    dec, attach, root = synth(B, L, seed, "cpu", torch.float32)
    md, ma = oracle.dmv1o_merge(dec.numpy(), attach.numpy(), root.numpy())
    lengths = np.full(B, L, dtype=np.int64)
    oracle.dmv1o(md[:8], ma[:8], lengths[:8], "log", np.float32)      # warm
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 400:
        t0 = time.perf_counter()
        oracle.dmv1o(md, ma, lengths, "log", np.float32)
        t_total += time.perf_counter() - t0
        reps += 1
    return {"value": B * reps / t_total, "unit": "sentences/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x (B={B}, L={L}) fp32 inside+outside, C oracle (oracle/vlg_oracle.c), "
                      f"{threads} OpenMP threads, {t_total:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256, help="sentences per GPU")
    ap.add_argument("--len", type=int, default=40, dest="L")
    ap.add_argument("--regions", type=int, default=36)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="storage type of the potentials")
    ap.add_argument("--ragged", action="store_true", help="random lengths instead of all = L")
    ap.add_argument("--grad-mb", type=float, default=0.0, help="pad the all-reduced gradient to this many MB")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-align", action="store_true", help="skip the secondary alignment measurement")
    args = ap.parse_args()

    from vlgae_amd import _C
    from vlgae_amd import dist as vdist
    from vlgae_amd.torch_struct import functional as F
    import vlgae_amd.torch_struct as ts

    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    # Debug aid for boxes with fewer GPUs than ranks (functional test of the N > 1 path only, never a
    # measurement): VLGAE_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo instead of RCCL.
    share = os.environ.get("VLGAE_BENCH_SHARE_GPU") == "1"
    rank, local_rank, world = vdist.init_from_env(backend="gloo" if share else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    lib = _C.lib()

    B, L = args.batch, args.L
    N = L + 1
    in_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    dec, attach, root = synth(B, L, 1000 + rank, dev, torch.float32)
    md32, ma32 = ts.DMV1o.merge(dec, attach, root)                    # merge is the scorer's job (ldndmv.py:209)
    md, ma = md32.to(in_dtype).contiguous(), ma32.to(in_dtype).contiguous()
    if args.ragged:
        lengths = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(7 + rank))
        lengths[0] = L
    else:
        lengths = torch.full((B,), L, dtype=torch.long)
    lengths_np = lengths.numpy().copy()
    lengths = lengths.to(dev)

    logZ = torch.empty(B, dtype=torch.float32, device=dev)
    gdec = torch.empty((B, N, 2, 2, 2), dtype=torch.float32, device=dev)
    gatt = torch.empty((B, N, N, 2), dtype=torch.float32, device=dev)
    ws_bytes = lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, B, N, 0)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    dt_code = _C.BF16 if in_dtype == torch.bfloat16 else _C.F32
    stream = torch.cuda.current_stream(dev)
    sp = _C.ctypes.c_void_p(stream.cuda_stream)
    p = [_C.ptr(x) for x in (md, ma, lengths, logZ, gdec, gatt, ws)]

    n_grad = N * 8 + N * N * 2
    pad = int(args.grad_mb * 1e6 / 4)
    reducer = vdist.GradAllReducer(max(n_grad, pad), dev) if world > 1 else None

    def step():
        rc = lib.vlg_dmv1o_inside_outside(p[0], p[1], p[2], B, N, dt_code, 0, None, p[3], p[4], p[5], p[6], ws_bytes, sp)
        if rc:
            _C.check(rc, "dmv1o_inside_outside")
        if reducer is not None:   # marginal-loss gradient of position-tied parameters = batch-summed counts
            buf = reducer.buffer
            torch.sum(gdec.view(B, -1), 0, out=buf[:N * 8])
            torch.sum(gatt.view(B, -1), 0, out=buf[N * 8:n_grad])
            reducer.launch()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        step()
    ev1.record(stream)
    if reducer is not None:
        reducer.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- checks outside the timed region: finite, and counts sum to the number of words ----
    assert bool(torch.isfinite(logZ).all()), "non-finite logZ"
    arcs = float(gatt.sum().item())
    assert abs(arcs - float(lengths_np.sum())) < 1e-3 * lengths_np.sum(), (arcs, lengths_np.sum())

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    sent_per_s = B * world * args.steps / elapsed
    kern_s = gpu_ms * 1e-3 / args.steps                   # HIP events on the launch stream, back-to-back launches
    alg_bytes = algorithmic_bytes(B, N, 2 if in_dtype == torch.bfloat16 else 4)
    achieved = alg_bytes / kern_s / 1e9
    exp_ops = exp_class_ops(lengths_np)
    exp_peak = N_CU * SIMD_PER_CU * TRANS_LANES_PER_CLK * CLOCK_GHZ * 1e9
    # HBM bytes per launch from the PMC counters: cannot be collected live (rocprofv3 --pmc runs in its own pass);
    # taken from the committed profile of THIS workload, with the gfx950 FETCH_SIZE x2 correction applied.
    traffic = None
    prof = os.path.join(ROOT, "profiles", "r01_g_pmc_traffic.json")
    if os.path.exists(prof) and (B, L, args.dtype, bool(args.ragged)) == (256, 40, "bf16", False):
        t = json.load(open(prof)).get("dmv1o_kernel")
        if t:
            traffic = (2.0 * t["FETCH_SIZE"]["avg_KB"] + t["WRITE_SIZE"]["avg_KB"]) * 1024.0
    out = {
        "metric": "sentences/sec, batched inside-outside L=%d B=%d" % (L, B),
        "value": sent_per_s, "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "DMV1o inside+outside (Log semiring) -> logZ + expected counts, "
                               f"B={B}/GPU L={L} N={N}, potentials stored {args.dtype}, fp32 charts; "
                               "BASELINE.json configs[1]" + (" sharded x%d, configs[2]" % world if world > 1 else ""),
                   "global_batch": B * world, "seq_len": L, "ragged": bool(args.ragged),
                   "parallelism": (f"dp{world}" if world > 1 else "single") + (" (DEBUG: ranks share one GPU, gloo)" if share else ""),
                   "allreduce_floats": (max(n_grad, pad) if world > 1 else 0)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "traffic_source": "profiles/r01_g_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH x2)" if traffic else None,
                     "kernel": "dmv1o_kernel<Log, mode 0 (all charts in LDS), fused inside+outside>", "kernel_us": kern_s * 1e6,
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "note": "latency-bound DP: 2(N-1) barrier-separated width steps per sentence, one workgroup "
                             "per sentence; see exp_rate for the bound that binds"},
        "exp_rate": {"achieved_Gops": exp_ops / kern_s / 1e9, "peak_Gops": exp_peak / 1e9,
                     "frac": exp_ops / kern_s / exp_peak, "ops_per_launch": exp_ops,
                     "note": "exp-class ops (N^3-N inside + 2x outside per sentence) vs v_exp_f32 issue peak "
                             "256 CU x 4 SIMD x 8 lanes/clk x 2.4 GHz"},
    }

    # ---- the same step through the drop-in Python API (DMV1o(...).partition + autograd.grad) ----
    d_, a_ = md.detach().requires_grad_(), ma.detach().requires_grad_()
    for _ in range(10):
        torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    n_api = 100
    for _ in range(n_api):
        torch.autograd.grad(ts.DMV1o([d_, a_], lengths).partition.sum(), [d_, a_])
    torch.cuda.synchronize(dev)
    out["api_path"] = {"sentences_per_s": B * n_api / (time.perf_counter() - t0),
                       "what": "DMV1o([dec,attach],lengths).partition.sum() + torch.autograd.grad, 1 GPU"}

    # ---- Viterbi decode of the same batch: Max-semiring inside + back-pointer walk -> head vector (joint.py:256-258) ----
    from vlgae_amd.torch_struct import functional as Fn
    for _ in range(5):
        Fn.dmv1o_decode(md, ma, lengths)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        Fn.dmv1o_decode(md, ma, lengths)
    e1.record()
    torch.cuda.synchronize(dev)
    out["decode"] = {"us": e0.elapsed_time(e1) / 50 * 1e3, "sentences_per_s": B * 50 / (e0.elapsed_time(e1) * 1e-3),
                     "what": "dmv1o_decode: best tree as heads [B,N], on device"}
    pair = lambda: ts.DMV1o([md, ma], lengths).marginals_and_heads()
    for _ in range(5):
        pair()
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(50):
        pair()
    e1.record()
    torch.cuda.synchronize(dev)
    out["marginals_and_heads"] = {"us": e0.elapsed_time(e1) / 50 * 1e3,
                                  "what": "arc marginals + Viterbi heads of one batch (joint.py:251-258), two HIP streams"}

    # ---- secondary: the region x word alignment that feeds / consumes the DP (joint.py:406-419) ----
    if not args.no_align and world == 1:   # single-GPU secondary measurements; multi-GPU runs report the headline only
        from vlgae_amd import align
        Q, V, d = 2 * N, args.regions, 128
        g = torch.Generator().manual_seed(5)
        txt = torch.randn(B, Q, d, generator=g).to(dev, in_dtype)
        vis = torch.randn(B, V, d, generator=g).to(dev, in_dtype)
        for full in (True, False):
            kw = dict(full=full, max_v=not full, max_q=not full)
            for _ in range(20):   # long enough for the clocks to settle after the idle gap before this section
                r = align.bilinear_align(txt, vis, **kw)
            torch.cuda.synchronize(dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n_al = 100
            for _ in range(n_al):
                r = align.bilinear_align(txt, vis, **kw)
            e1.record()
            torch.cuda.synchronize(dev)
            del r
            sec = e0.elapsed_time(e1) * 1e-3 / n_al
            flops = 2.0 * B * B * Q * V * d
            esz = 2 if in_dtype == torch.bfloat16 else 4
            byts = (B * Q + B * V) * d * esz + (B * B * Q * V * 4 if full else (B * B * (Q + V)) * 4)
            out["align_full" if full else "align_fused_max"] = {
                "sentences_per_s": B / sec, "ms": sec * 1e3, "TFLOP/s": flops / sec / 1e12,
                "GB/s": byts / sec / 1e9, "frac_hbm": byts / sec / 1e9 / HBM_PEAK_GBS,
                "shape": f"B=A={B} Q={Q} V={V} d={d} {args.dtype} in, fp32 out"}

        # the two consumers on the training path: attention-fuse (joint.py:670-674) and the grounding loss on the
        # fused maxima (joint.py:439-491), forward + gradients, through the host API
        def timed(fn, n):
            """ms per call: median of three windows of n calls (a host hiccup inside one window -- an allocator refill,
            a scheduler tick -- otherwise lands in a host-bound entry as a 10x outlier)."""
            for _ in range(10):
                fn()
            wins = []
            for _ in range(3):
                torch.cuda.synchronize(dev)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize(dev)
                wins.append(e0.elapsed_time(e1) / n)
            return sorted(wins)[1]
        h = 256
        mk = lambda *shape: torch.randn(*shape, generator=g).to(dev, in_dtype).requires_grad_(True)
        f_vis, f_txt, f_mid, f_enc = mk(B, V, d), mk(B, N, d), mk(B, V, h), mk(B, L, h)
        ln_w, ln_b = torch.ones(h, device=dev, requires_grad=True), torch.zeros(h, device=dev, requires_grad=True)
        dout = torch.randn(B, L, h, generator=g).to(dev)
        leaves = [f_vis, f_txt, f_mid, f_enc, ln_w, ln_b]
        out["attention_fuse"] = {
            "fwd_ms": timed(lambda: align.attention_fuse(f_vis.detach(), f_txt.detach(), f_mid.detach(), f_enc.detach(),
                                                         ln_w.detach(), ln_b.detach(), 1e-5), 50),
            "fwd_bwd_ms": timed(lambda: torch.autograd.grad(align.attention_fuse(*leaves, 1e-5), leaves, dout), 50),
            "shape": f"B={B} L={L} V={V} d={d} h={h} {args.dtype} in; host API incl. autograd overhead"}
        tmask = torch.ones(B, Q, dtype=torch.bool, device=dev)
        tmask[:, 0] = tmask[:, N] = False
        vmask = torch.ones(B, V, dtype=torch.bool, device=dev)
        marg = torch.rand(B, Q, generator=g).to(dev) * tmask
        g_txt, g_vis = mk(B, Q, d), mk(B, V, d)
        def ground():
            total, _ = align.grounding_loss_factor_ce(g_txt, g_vis, tmask, vmask, marg, B * L, 1.0)
            return torch.autograd.grad(total, [g_txt, g_vis])
        out["grounding_loss"] = {"fwd_bwd_ms": timed(ground, 10),
                                 "shape": f"B=A={B} Q={Q} V={V} d={d} {args.dtype} in; loss + gradients, no [B,A,Q,V] tensor"}

        out["grounding_decode"] = {
            "ms": timed(lambda: align.grounding_decode(g_txt.detach(), g_vis.detach(), tmask, vmask), 20),
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
            "grounding_loss_fwd_bwd_ms": timed(ground_shipped, 10),
            "grounding_decode_ms": timed(lambda: align.grounding_decode(s_txt.detach(), s_vis.detach(), s_tmask, s_vmask), 10),
            "shape": f"B=A={Bs} Q={Q} V={Vs} d={d} {args.dtype} in"}
        del s_txt, s_vis

        # arc encoder's trilinear term (joint.py:282-284): M = B * (L + 1) rows, 128^3 weights
        a_child, a_parent = mk(B, N, d), mk(B, N, d)
        a_w1 = (torch.randn(d, d, d, generator=g) / d).to(dev, in_dtype).requires_grad_(True)
        a_dout = torch.randn(B, N, d, generator=g).to(dev)
        out["arc_trilinear"] = {
            "fwd_ms": timed(lambda: align.arc_trilinear(a_child.detach(), a_w1.detach(), a_parent.detach()), 20),
            "fwd_bwd_ms": timed(lambda: torch.autograd.grad(align.arc_trilinear(a_child, a_w1, a_parent),
                                                            [a_child, a_w1, a_parent], a_dout), 10),
            "shape": f"M={B * N} X=H=Y={d} {args.dtype} in; einsum('bcx,xhy,bcy->bch') without the [M,H,Y] intermediate"}

    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(B, L, 1000, args.cpu_seconds)
        out["speedup_vs_cpu_baseline"] = sent_per_s / out["cpu_baseline"]["value"]
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
