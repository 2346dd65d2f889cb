"""bench.py --workload train_step: one training step (BASELINE.json configs[4], vlgae_amd/train_step.py in the reference's wiring, from
the frozen features) sharded data-parallel.

Every rank builds its own shard of the step (B sentences / GPU, rank-specific synthetic batch -- the
`ConstantTokenNumSampler(rank=, world_size=)` arrangement: rank r takes batches r, r+W, ...), runs forward + backward and all-reduces the
flat fp32 gradient of EVERY trainable parameter of the VLGAE model (~7 M fp32 = 28 MB, SURVEY.md 8e: the text encoder, the visual encoder's
box_fc / rel_fc / attr_fc, vis_mlp_pre_matching, LayerNorm, word | child | parent encoders, arc encoder, the parser's feed-forwards and
scorers) over RCCL with synchronous-SGD semantics and DDP's averaging -- every bucket is complete (stream-wise) before the next step's
first kernel, as under Lightning DDP (/root/reference config/trainer/train.yaml:27-29, src/pipeline.py:112-126).  Every float of the
buffer is a gradient the step computed (round 5: the step starts at the frozen features, so no stand-in filler is left); the gradients
sit in `--buckets` pieces in the order they become final during the backward pass (train_step `ready_groups`), and a piece starts
reducing from a leaf hook the moment its last gradient exists, overlapping the adjoints still to run.  At the object-factor-only layout
of BASELINE.json configs[1] (R = 36 region columns) the model's rel_fc / attr_fc receive no gradient -- they ride along as the zeros DDP
all-reduces for unused parameters (`find_unused_parameters`), so that the collective has the shipped model's size at either layout.

Reported: value (sentences/s, whole job), step_ms (with the collectives), compute_ms (same step, no collective),
allreduce_ms (the pieces alone, back to back), overlap_frac = 1 - (step_ms - compute_ms) / allreduce_ms.
"""
import time

import torch
import torch.distributed as dist

FF_SHAPES = dict(E=800, h=256, Et=32, T=45, H=256, nb=150, r=16)   # the parser's feed-forwards at the shipped widths (vlgae.yaml)
N_VIS = 2048                                                        # Faster-RCNN region feature width (vis_encoder.n_in, vlgae.yaml:29)
MODEL_VIS_ENCODERS = 3                                              # box_fc, rel_fc, attr_fc of the shipped model (use_attr: true)


def _param_shapes(d=128, h=256, n_enc=1):
    """name -> shape of every trainable leaf of train_step.build(wiring="reference") with n_enc visual-encoder MLPs on the path, and the
    readiness groups (no GPU needed)."""
    f = FF_SHAPES
    shapes = dict(b=(d,), b_enc=(3 * d,), ln_b=(h,), ln_w=(h,), w1=(d, d, d), w2=(d, d), w_enc=(3 * d, h), w_vis=(d, h),
                  w_text=(h, f["E"]), w_venc=(n_enc * h, 2 * N_VIS), b_venc=(n_enc * h,),
                  token_emb=(f["T"], f["Et"]), root_emb=(1, 10), dec_emb=(2, 10))

    def lin(name, n_in, n_out):
        shapes[name + ".weight"], shapes[name + ".bias"] = (n_out, n_in), (n_out,)
    for name, n_in in (("head_ff", f["E"] + f["h"]), ("child_ff", f["Et"]), ("root_ff", 10), ("dec_ff", 10)):
        lin(f"ff.{name}.linear", n_in, f["H"])
    for name in ("HASCHILD_linear", "NOCHILD_linear", "LEFT_linear", "RIGHT_linear"):
        lin(f"ff.mid_ff.{name}.0", f["H"], f["nb"])
        lin(f"ff.mid_ff.{name}.1", f["nb"], f["H"])
    for name in ("valence_linear", "direction_linear", "linear1", "linear2"):
        lin(f"ff.mid_ff.{name}", f["H"], f["H"])
    for name in ("attach_scorer", "dec_scorer", "root_scorer"):
        lin(f"ff.{name}.project1", f["H"], f["r"])
        lin(f"ff.{name}.project2", f["H"], f["r"])
    ff_names = sorted(k for k in shapes if k.startswith("ff.") or k in ("token_emb", "root_emb", "dec_emb"))
    return shapes, (["w1", "w2", "b"], ff_names, ["ln_w", "ln_b", "w_enc", "b_enc", "w_vis"], ["w_text", "w_venc", "b_venc"])


class _DryStep:
    """CPU stand-in for the launcher / collective plumbing tests (VLGAE_BENCH_DRYRUN=1): no kernels, fixed fake gradients handed to
    `on_grad` in the readiness order of the real step."""

    def __init__(self, B, L):
        self.shapes, self.ready_groups = _param_shapes()
        self.grads = {k: torch.ones(s) for k, s in self.shapes.items()}
        self.lengths = torch.full((B,), L, dtype=torch.long)
        self.trainable = tuple(sorted(self.shapes))
        self.on_grad = None

    def __call__(self, stage_hook=None):
        if stage_hook is not None:
            stage_hook()
        for group in self.ready_groups:
            for k in group:
                if self.on_grad is not None:
                    self.on_grad(k, self.grads[k])
        return torch.zeros(()), self.grads, ()


def measure(args, rank, world, dev, dry, barrier):
    """Runs the sharded train step; returns the result dict on every rank (rank 0 prints it).  Backward runs on the calling
    thread (one process per GPU has no use for torch's per-device engine thread): the bucket hooks then
    issue their collectives from the same thread, on the same current stream, as every kernel of the step."""
    with torch.autograd.set_multithreading_enabled(False):
        return _measure(args, rank, world, dev, dry, barrier)


def _all_ranks_ok(ok, world, dev, dry):
    """Every rank learns whether EVERY rank got this far, before any timed collective: a rank that failed to build its step would
    otherwise leave the others hanging inside an all-reduce."""
    if world <= 1:
        return ok
    flag = torch.tensor([1.0 if ok else 0.0], device="cpu" if dry else dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() > 0.5)


def _measure(args, rank, world, dev, dry, barrier):
    from vlgae_amd import dist as vdist
    from vlgae_amd import train_step
    B, L, V = args.batch, args.L, args.regions
    factors = tuple(getattr(args, "factors", ()) or ())
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    step, err = None, None
    try:
        if dry:
            step = _DryStep(B, L)
        else:
            step = train_step.build(B, L, V, dev, dtype=dtype, seed=11 + rank, factors=factors, n_vis=N_VIS, **FF_SHAPES)
            step()                                         # one untimed step: every kernel and allocation path exercised
    except Exception as e:                                 # noqa: BLE001 -- reported on every rank below
        err = e
    if not _all_ranks_ok(err is None, world, dev, dry):
        raise RuntimeError(f"train_step could not be built on every rank (this rank: {err!r})")
    shapes = {k: tuple(step.P[k].shape) for k in step.trainable} if not dry else step.shapes
    # ---- the flat gradient: [ group 0 | group 1 | ... | slot 0 = word count | last group | filler up to the model's size ] ----
    n_b = max(1, min(args.buckets, len(step.ready_groups)))
    groups = [list(gp) for gp in step.ready_groups[:n_b - 1]] + [[k for gp in step.ready_groups[n_b - 1:] for k in gp]]
    numel = lambda k: int(torch.Size(shapes[k]).numel())
    offsets, bounds, o = {}, [], 0
    for gi, gp in enumerate(groups):
        lo = o
        if gi == len(groups) - 1:
            o += 1                                          # slot `check`: this rank's word count (the mean-over-ranks check)
        for k in gp:
            offsets[k] = o
            o += numel(k)
        bounds.append([lo, o])
    n_real = o - 1
    # the visual-encoder MLPs of the shipped model that this factor layout does not reach (rel_fc / attr_fc at the object-only layout):
    # no gradient, all-reduced as zeros like DDP's unused parameters -- the buffer has the model's size at either layout
    n_enc = shapes["w_venc"][0] // shapes["w_text"][0]
    n_unused = (MODEL_VIS_ENCODERS - n_enc) * (shapes["w_venc"][0] // n_enc) * (shapes["w_venc"][1] + 1)
    n_model = o + max(0, n_unused)
    bounds[-1][1] = n_model
    red = vdist.BucketedGradReducer(n_model, dev, bounds=[tuple(b) for b in bounds], average=True)
    check = bounds[-1][0]
    views = {k: red.flat[off:off + numel(k)] for k, off in offsets.items()}
    group_of = {k: gi for gi, gp in enumerate(groups) for k in gp}
    words_local = float(step.lengths.sum().item())
    sync = (lambda: None) if dry else (lambda: torch.cuda.synchronize(dev))
    pending = [0] * len(groups)
    state = {"comm": False}

    def on_grad(name, g):                                  # inside the backward pass: the gradient of `name` is final
        views[name].copy_(g.reshape(-1))                   # cast to fp32 in the copy
        gi = group_of[name]
        pending[gi] -= 1
        if pending[gi] == 0 and state["comm"]:
            if gi == len(groups) - 1:
                red.flat[check:check + 1].fill_(words_local)
            red.launch(gi)
    step.on_grad = on_grad

    def run_eager(comm):
        state["comm"] = comm
        for gi, gp in enumerate(groups):
            pending[gi] = len(gp)
        if comm:
            red.wait()                                     # synchronous SGD: last step's reduced gradient before this step's first kernel
        step(None)
        assert not any(pending), pending                   # every parameter's gradient arrived through its hook

    # step_mode "graph" (default on a GPU): forward + backward + the packing of every gradient into the flat buffer are captured ONCE as a
    # HIP graph; a step is one replay, and the buckets' collectives are enqueued behind it.  Eager, the step is host-bound (~2.6 ms of
    # Python / autograd enqueue for ~1.95 ms of device work at B = 256): ranks would run at the speed of their interpreters and drift
    # apart between collectives.  What the graph form gives up is the overlap of a bucket's all-reduce with the adjoints still to run
    # (the leaf-hook launches of the eager form); with ~26 MB per step over xGMI that is ~0.2 ms against the 0.6 ms the host costs.
    mode = "eager" if dry else getattr(args, "step_mode", "graph")
    graph = None
    if mode == "graph":
        state["comm"] = False
        for gi, gp in enumerate(groups):
            pending[gi] = len(gp)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                run_eager(False)
        torch.cuda.current_stream(dev).wait_stream(side)
        # Capture with a process group alive (its communicator was created eagerly in dist.warm_up, before this point; the watchdog
        # thread only polls work objects of collectives, none of which is in flight here).  capture_error_mode "thread_local": calls
        # other threads make while this one captures (the watchdog's event queries) do not invalidate the capture.  A capture that
        # throws anyway must not cost the line: fall back to the eager step in-process and report the mode that actually ran.
        for gi, gp in enumerate(groups):
            pending[gi] = len(gp)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                step(None)
                red.flat[check:check + 1].fill_(words_local)   # (the averaged slot of the previous step is overwritten every replay)
            assert not any(pending), pending
        except Exception as e:                                 # noqa: BLE001 -- any capture failure takes the same exit
            graph, mode = None, "eager"
            state["capture_error"] = repr(e)[:300]
            torch.cuda.synchronize(dev)

    def run(comm):
        if graph is None:
            return run_eager(comm)
        if comm:
            red.wait()                                     # synchronous SGD: last step's reduced gradient before this step's first kernel
        graph.replay()
        if comm:
            for gi in range(len(groups)):                  # stream-ordered behind the replay; in readiness order
                red.launch(gi)

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

    el_compute = timed(lambda: run(False))
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
        el = timed(lambda: run(True), tail=red.wait)
        sync()
        # the reduced buffer must be the MEAN over ranks (DDP's semantics): the check slot carries every rank's word count
        tot = torch.tensor([words_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tot)
        got = float(red.flat[check].item())
        want = float(tot.item()) / world
        assert abs(got - want) <= 1e-3 * want, (got, want)
        step_ms, compute_ms, ar_ms = el * 1e3 / args.steps, el_compute * 1e3 / args.steps, ar_s * 1e3
        res["comm"] = {"backend": dist.get_backend(), "rccl_ranks_seen": vdist.warm_up(dev),
                       "allreduce_ms": ar_ms, "allreduce_bytes": n_model * 4, "buckets": red.n_buckets,
                       "bucket_bounds": red.bounds, "bucket_contents": [gp if len(gp) <= 6 else gp[:3] + [f"... {len(gp) - 3} more"] for gp in groups],
                       "allreduce_busbw_GBs": 2.0 * (world - 1) / world * n_model * 4 / ar_s / 1e9,
                       "overlap_frac": max(0.0, min(1.0, 1.0 - (step_ms - compute_ms) / ar_ms)) if ar_ms > 0 else None,
                       "mean_over_ranks_check": {"slot": got, "expected": want},
                       "semantics": "synchronous SGD, gradients AVERAGED over ranks (DDP): every bucket of step k is reduced "
                                    "(stream-ordered) before step k+1's first kernel; " + ("the buckets are enqueued behind the replayed step graph, in readiness order"
                                    if mode == "graph" else "a bucket starts from inside the backward pass the moment the last of ITS parameters' gradients exists (leaf hooks)"),
                       "payload": f"{n_real} gradient floats, one per trainable parameter the step reaches (text encoder, visual encoder, "
                                  "vis_mlp_pre_matching, LayerNorm, word | child | parent encoders, arc encoder, parser feed-forwards and scorers), "
                                  f"in readiness order; + {n_model - n_real - 1} zeros for the shipped model's visual-encoder MLPs this factor "
                                  "layout does not reach (DDP all-reduces unused parameters as zeros); no filler"}
    else:
        el = el_compute
        step_ms = compute_ms = el * 1e3 / args.steps
    res.update(value=B * world * args.steps / el, step_ms=step_ms, compute_ms=compute_ms, step_mode=mode,
               real_gradient_floats=n_real, allreduce_floats=n_model if world > 1 else 0)
    if "capture_error" in state:
        res["capture_error"] = state["capture_error"]          # graph mode was asked for and fell back to eager
    return res


def json_line(args, world, res, dry, share=False):
    B, L, V = args.batch, args.L, args.regions
    out = {"metric": "sentences/sec, full train step hot path (configs[4]) L=%d B=%d/GPU" % (L, B),
           "value": res["value"], "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": res["step_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype,
           "data": "synthetic" if not dry else "DRY RUN: no kernels ran (launcher / collective plumbing on CPU, gloo); not a measurement",
           "config": {"workload": "one training step as the reference wires it (vlgae_amd/train_step.py, pinned on reference-made fixtures), from "
                                  "the frozen features: text encoder (dropout + Linear 800->256) + visual encoder (box_fc [; rel_fc; attr_fc] on "
                                  "2048-d region features) -> attention fuse -> parser feed-forwards -> score construction -> DMV1o marginals + "
                                  "Viterbi heads -> word | child | parent encoders + arc encoder -> alignment maxima + grounding cross-entropy -> "
                                  f"-DMV1o.max -> 0.5/0.5 -> gradients of every trainable parameter; B={B}/GPU L={L} R={V} d=128 h=256, "
                                  f"{args.dtype} storage, synthetic frozen features (random 800-d embeddings / 2048-d region features); BASELINE.json configs[4]",
                      "global_batch": B * world, "seq_len": L,
                      "parallelism": (f"dp{world}" if world > 1 else "single") + (" (DEBUG: ranks share one GPU, gloo)" if share else ""),
                      "allreduce_floats": res["allreduce_floats"]},
           "compute_ms": res["compute_ms"], "step_ms": res["step_ms"], "real_gradient_floats": res["real_gradient_floats"],
           "step_mode": res.get("step_mode", "eager") + (": forward + backward + gradient packing replayed as ONE captured HIP graph, the buckets' collectives "
                                                         "enqueued behind it" if res.get("step_mode") == "graph" else
                                                         ": Python / autograd enqueue per step, a bucket's collective launched from a leaf hook inside the backward pass")}
    if "comm" in res:
        out["comm"] = res["comm"]
    if "capture_error" in res:
        out["capture_error"] = res["capture_error"]
    if dry:
        out["dry_run"] = True
    return out
