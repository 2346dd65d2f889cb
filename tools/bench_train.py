"""bench.py --workload train_step: the chained training-step hot path (BASELINE.json configs[4]) sharded data-parallel.

Every rank builds its own shard of the step (tools/train_step.py: B sentences / GPU, rank-specific synthetic batch --
the `ConstantTokenNumSampler(rank=, world_size=)` arrangement: rank r takes batches r, r+W, ...), runs forward +
backward, packs the step's real leaf gradients of the trainable weights at the head of a flat fp32 buffer padded to the
VLGAE model's gradient (--grad-mb, ~7 M fp32 = 28 MB) and all-reduces it over RCCL with synchronous-SGD
semantics -- every bucket is complete (stream-wise) before the next step's first kernel, as under Lightning DDP
(/root/reference config/trainer/train.yaml:27-29, src/pipeline.py:112-126).  The buffer goes in `--buckets` pieces:
the tail piece(s) -- the stand-in for parameters whose gradients are complete once the DP and grounding-loss adjoints
have run (the scorer / matching heads: DDP's first buckets) -- start reducing from a hook inside the backward pass and
overlap the arc-encoder / projection / attention-fuse adjoints; the head piece (the real leaf gradients) goes last.

Reported: value (sentences/s, whole job), step_ms (with the collective), compute_ms (same step, no collective),
allreduce_ms (the buckets alone, back to back), overlap_frac = 1 - (step_ms - compute_ms) / allreduce_ms.
"""
import os
import sys
import time

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

TRAINABLE = ("b", "b_enc", "ln_b", "ln_w", "w1", "w2", "w_enc")   # leaves of train_step.build that are parameters


class _DryStep:
    """CPU stand-in for the launcher / collective plumbing tests (VLGAE_BENCH_DRYRUN=1): no kernels, fixed fake gradients."""

    def __init__(self, B, L, d=128, h=256):
        shapes = dict(b=(d,), b_enc=(3 * d,), ln_b=(h,), ln_w=(h,), w1=(d, d, d), w2=(d, d), w_enc=(3 * d, h))
        self.grads = {k: torch.ones(s) for k, s in shapes.items()}
        self.lengths = torch.full((B,), L, dtype=torch.long)

    def __call__(self, stage_hook=None):
        if stage_hook is not None:
            stage_hook()
        return torch.zeros(()), self.grads, ()


def measure(args, rank, world, dev, dry, barrier):
    """Runs the sharded train step; returns the result dict on every rank (rank 0 prints it).  Backward runs on the calling
    thread (one process per GPU has no use for torch's per-device engine thread): the bucket hook then
    issues its collective from the same thread, on the same current stream, as every kernel of the step."""
    with torch.autograd.set_multithreading_enabled(False):
        return _measure(args, rank, world, dev, dry, barrier)


def _measure(args, rank, world, dev, dry, barrier):
    from vlgae_amd import dist as vdist
    import train_step
    B, L, V = args.batch, args.L, args.regions
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    if dry:
        step = _DryStep(B, L)
    else:
        step = train_step.build(B, L, V, dev, dtype=dtype, seed=11 + rank)
    n_real = sum(int(torch.Size(s).numel()) for s in
                 (dict(b=(128,), b_enc=(384,), ln_b=(256,), ln_w=(256,), w1=(128, 128, 128), w2=(128, 128), w_enc=(384, 256)).values()))
    head = 1 + n_real                                  # slot 0: this rank's word count (the sum-over-ranks check)
    n_model = max(head, int(args.grad_mb * 1e6 / 4))
    red = vdist.BucketedGradReducer(n_model, dev, n_buckets=args.buckets, head=head)
    words_local = float(step.lengths.sum().item())
    head_view = red.head_view
    sync = (lambda: None) if dry else (lambda: torch.cuda.synchronize(dev))

    def early():                                       # inside the backward pass: start the tail bucket(s)
        for i in range(red.n_buckets - 1):
            red.launch(i)

    def pack(grads):
        head_view[0:1].fill_(words_local)
        o = 1
        for k in TRAINABLE:
            g = grads[k]
            n = g.numel()
            head_view[o:o + n].copy_(g.reshape(-1))    # cast to fp32 in the copy
            o += n

    def step_comm():
        red.wait()                                     # synchronous SGD: last step's reduced gradient before this step's first kernel
        _, grads, _ = step(early)
        pack(grads)
        red.launch(red.n_buckets - 1)

    def step_compute():
        _, grads, _ = step(None)
        pack(grads)

    def timed(fn, tail=None):
        for _ in range(args.warmup):
            fn()
        if tail:
            tail()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fn()
        if tail:
            tail()
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    el_compute = timed(step_compute)
    res = {}
    if world > 1:
        def ar_only():
            for i in range(red.n_buckets):
                red.launch(i)
            red.wait()
        for _ in range(3):
            ar_only()
        barrier()
        t0 = time.perf_counter()
        n_ar = 10
        for _ in range(n_ar):
            ar_only()
        barrier()
        ar_s = (time.perf_counter() - t0) / n_ar
        red.flat.zero_()
        el = timed(step_comm, tail=red.wait)
        sync()
        # the reduced buffer must be the sum over ranks: slot 0 carries every rank's word count
        tot = torch.tensor([words_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        got = float(red.head_view[0].item())
        assert abs(got - float(tot.item())) <= 1e-3 * float(tot.item()), (got, float(tot.item()))
        step_ms, compute_ms, ar_ms = el * 1e3 / args.steps, el_compute * 1e3 / args.steps, ar_s * 1e3
        res["comm"] = {"backend": dist.get_backend(), "rccl_ranks_seen": vdist.warm_up(dev),
                       "allreduce_ms": ar_ms, "allreduce_bytes": n_model * 4, "buckets": red.n_buckets,
                       "bucket_bounds": red.bounds,
                       "allreduce_busbw_GBs": 2.0 * (world - 1) / world * n_model * 4 / ar_s / 1e9,
                       "overlap_frac": max(0.0, min(1.0, 1.0 - (step_ms - compute_ms) / ar_ms)) if ar_ms > 0 else None,
                       "sum_over_ranks_check": {"slot0": got, "expected": float(tot.item())},
                       "semantics": "synchronous SGD: every bucket of step k is reduced (stream-ordered) before step k+1's "
                                    "first kernel; tail bucket(s) start from a hook inside the backward pass"}
    else:
        el = el_compute
        step_ms = compute_ms = el * 1e3 / args.steps
    res.update(value=B * world * args.steps / el, step_ms=step_ms, compute_ms=compute_ms,
               real_gradient_floats=n_real, allreduce_floats=n_model if world > 1 else 0)
    return res


def json_line(args, world, res, dry, share=False):
    B, L, V = args.batch, args.L, args.regions
    out = {"metric": "sentences/sec, full train step hot path (configs[4]) L=%d B=%d/GPU" % (L, B),
           "value": res["value"], "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": res["step_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype,
           "data": "synthetic" if not dry else "DRY RUN: no kernels ran (launcher / collective plumbing on CPU, gloo); not a measurement",
           "config": {"workload": "attention-fuse -> projections -> DMV1o marginals + Viterbi heads -> arc encoder -> alignment "
                                  "maxima + grounding cross-entropy -> -DMV1o.max -> gradients (tools/train_step.py), "
                                  f"B={B}/GPU L={L} V={V} d=128 h=256, {args.dtype} features, synthetic encoder outputs; "
                                  "BASELINE.json configs[4]",
                      "global_batch": B * world, "seq_len": L,
                      "parallelism": (f"dp{world}" if world > 1 else "single") + (" (DEBUG: ranks share one GPU, gloo)" if share else ""),
                      "allreduce_floats": res["allreduce_floats"]},
           "compute_ms": res["compute_ms"], "step_ms": res["step_ms"]}
    if "comm" in res:
        out["comm"] = res["comm"]
    if dry:
        out["dry_run"] = True
    return out
