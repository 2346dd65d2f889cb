#!/usr/bin/env python3
"""bench.py -- sentences/sec of the batched inside-outside hot path on MI355X.

    python bench.py                          # 1 GPU, B=256 L=40 (BASELINE.json configs[1])
    python bench.py --gpus 8                 # starts 8 fresh rank processes itself (torch.distributed.run), relays rank 0
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          # the driver's form: ranks already exist

A "step" is one pass of the hot path over one batch of synthetic root-merged potentials already resident
in HBM: the fused DMV1o inside+outside launch (Log semiring) producing logZ [B] and the expected counts
grad_dec [B,N,2,2,2] / grad_attach [B,N,N,2] -- what `torch.autograd.grad(DMV1o(...).partition.sum(), ...)`
costs in the reference (src/model/joint.py:254-255).  With N > 1 ranks every rank owns 256 sentences (weak
scaling, global batch 256*N, configs[2]) and each step ends with ONE RCCL all-reduce of the gradient: the
batch-summed expected counts at the head of a flat fp32 buffer padded to the size of the VLGAE model's
gradient (--grad-mb, default 25.9 MB = the 6.48 M trainable floats of the shipped model's path, the buffer
`--workload train_step` fills with real gradients), issued asynchronously so it overlaps the next step's
kernel; all of them complete inside the timed region.  `value` is measured with that model-sized
collective; `value_dp_grad_only` repeats the timed region with only the DP's own 14.8 KB gradient.

Rank 0 prints ONE JSON line (see README / DESIGN.md for the fields).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# HSA reads this once, when the first HIP call initialises the runtime: it must be in the environment before that
# (the host driver only supports dmabuf IPC; without it RCCL fails with hipIpcGetMemHandle: invalid argument).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
N_CU, SIMD_PER_CU = 256, 4
TRANS_LANES_PER_CLK = 8        # v_exp_f32 / v_log_f32: 8 cycles per wave64 instruction per SIMD -> 8 lanes/clk
CLOCK_GHZ = 2.4
N_REGIONS = 5                  # timed regions per headline figure: value = the first, value_spread = min / median / max of all
PREWARM = 200                  # untimed launches before the first warmup / timed region (~14 ms)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--batch", type=int, default=256, help="sentences per GPU")
    ap.add_argument("--len", type=int, default=40, dest="L")
    ap.add_argument("--regions", type=int, default=36)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"], help="storage type of the potentials")
    ap.add_argument("--ragged", action="store_true", help="random lengths instead of all = L")
    ap.add_argument("--grad-mb", type=float, default=25.92,
                    help="dp workload: size of the all-reduced flat gradient in MB (default: the 6.48 M trainable fp32 parameters of the shipped model's path "
                         "-- what --workload train_step all-reduces as real gradients)")
    ap.add_argument("--factors", nargs="*", default=[], choices=["rel", "attr", "img"],
                    help="train_step workload: visual factors beside the objects (shipped model: rel attr img -> 1369 columns at 36 regions)")
    ap.add_argument("--workload", default="dp", choices=["dp", "train_step"],
                    help="dp: the headline DMV1o inside+outside step (BASELINE.json metric); train_step: the chained "
                         "training-step hot path of configs[4], sharded data-parallel (vlgae_amd/bench/sharded_step.py)")
    ap.add_argument("--buckets", type=int, default=4, help="train_step: pieces the flat gradient is all-reduced in")
    ap.add_argument("--step-mode", default="graph", choices=["graph", "eager"], dest="step_mode",
                    help="train_step: replay the captured step as one HIP graph and enqueue the collectives behind it (default), or run it eagerly "
                         "with the collectives launched from leaf hooks inside the backward pass")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-secondary", "--no-align", action="store_true", dest="no_secondary",
                    help="headline only: skip the secondary single-GPU measurements")
    return ap.parse_args(argv)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes and relay rank 0's JSON line.

    Runs BEFORE torch is imported: this parent never touches the GPU (no HIP call, not even a device count), it
    only spawns `python -m torch.distributed.run ... bench.py <same args>` as a child, forwards the child's JSON
    line(s) to stdout and everything else to stderr, and exits with the child's return code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    for line in proc.stdout:
        is_json = line.lstrip().startswith("{") and '"metric"' in line
        (sys.stdout if is_json else sys.stderr).write(line)
        (sys.stdout if is_json else sys.stderr).flush()
    return proc.wait()


def synth(B, L, seed, device, dtype):
    """Potentials shaped like the scorer's output (normalised log-probs), SURVEY.md section 8d."""
    import torch
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
    import numpy as np
    n = np.asarray(lengths, dtype=np.float64) + 1
    return float((3 * (n ** 3 - n)).sum())


def cpu_model_string():
    """The host CPU's model name (SURVEY.md 8d asks for it beside the core count)."""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine() or "unknown"


def cpu_baseline(B, L, seed, budget_s, one_thread=True):
    """The CPU oracle (C restatement of the reference algorithm, fp32, OpenMP over sentences) timed on this
    box's host cores on a bounded sample of the same workload.  BASELINE.md section 4 relates it to the
    reference's own PyTorch-CPU path (measured side by side in the build container).  SURVEY.md 8(d): the thread
    count, the CPU model string and a 1-thread figure are reported with it."""
    import numpy as np
    import torch
    import oracle
    oracle.build()
    threads = oracle.max_threads()
    Bc = max(B, 16 * threads)          # enough sentences per core for the OpenMP loop to scale
    dec, attach, root = synth(Bc, L, seed, "cpu", torch.float32)
    md, ma = oracle.dmv1o_merge(dec.numpy(), attach.numpy(), root.numpy())
    lengths = np.full(Bc, L, dtype=np.int64)
    oracle.dmv1o(md[:8], ma[:8], lengths[:8], "log", np.float32)      # warm

    def timed(n_sent, budget, max_reps):
        reps, t_total = 0, 0.0
        while t_total < budget and reps < max_reps:
            t0 = time.perf_counter()
            oracle.dmv1o(md[:n_sent], ma[:n_sent], lengths[:n_sent], "log", np.float32)
            t_total += time.perf_counter() - t0
            reps += 1
        return reps, t_total

    share_1t = 0.25 if one_thread else 0.0
    reps, t_total = timed(Bc, budget_s * (1.0 - share_1t), 400)
    res = {"value": Bc * reps / t_total, "unit": "sentences/s", "cores": threads, "kind": "port",
           "cpu_model": cpu_model_string(),
           "sample": f"{reps} x (B={Bc}, L={L}) fp32 inside+outside, C oracle (oracle/vlg_oracle.c), "
                     f"{threads} OpenMP threads, {t_total:.1f} s",
           "reference_equivalent_note": "the reference's own PyTorch-CPU path ran 0.08-0.12x of this port on the same "
                                        "8 threads in the build container (BASELINE.md section 4)"}
    if one_thread:
        B1 = min(Bc, 64)                # a 1-thread pass over 64 sentences is ~0.1 s at L=40
        oracle.set_threads(1)
        try:
            reps1, t1 = timed(B1, budget_s * share_1t, 200)
        finally:
            oracle.set_threads(threads)
        res["value_1_thread"] = B1 * reps1 / t1
        res["sample_1_thread"] = f"{reps1} x (B={B1}, L={L}), 1 OpenMP thread, {t1:.1f} s"
    return res


def kernel_source_id():
    """Short hash of the DP kernel sources: committed PMC profiles carry it so stale traffic figures are not reused."""
    import hashlib
    h = hashlib.sha1()
    for f in ("vlg_dp_kernels.h", "vlg_dp_core.h"):
        h.update(open(os.path.join(ROOT, "vlgae_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:12]


def pmc_traffic(workload_key):
    """HBM bytes per launch from the PMC counters.  They cannot be collected live (rocprofv3 --pmc runs in its own
    passes), so the figure comes from the newest committed profile of THIS workload -- FETCH_SIZE x2 (gfx950
    correction) + WRITE_SIZE -- and is labelled with the kernel-source hash it was taken at."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        t = d.get(workload_key) or (d.get("dmv1o_kernel") if workload_key == "dmv1o_B256_L40_bf16" else None)
        if t and "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            best = (path, d, t)
    if best is None:
        return None, None
    path, d, t = best
    traffic = (2.0 * t["FETCH_SIZE"]["avg_KB"] + t["WRITE_SIZE"]["avg_KB"]) * 1024.0
    same = d.get("kernel_source_id") == kernel_source_id()
    src = (f"{os.path.relpath(path, ROOT)} (committed rocprofv3 --pmc passes FETCH_SIZE / WRITE_SIZE, FETCH x2; "
           f"profiled at kernel source {d.get('kernel_source_id', 'r01')}, "
           f"{'same as' if same else 'DIFFERENT from'} this build {kernel_source_id()} -- the byte traffic is set by the "
           "I/O stage, which only moves when the load/store policy changes)")
    return traffic, src


class Headline:
    """Buffers and launch closure of the headline workload on one rank."""

    def __init__(self, args, rank, dev, dry):
        import numpy as np
        import torch
        self.args, self.dev, self.dry = args, dev, dry
        B, L = args.batch, args.L
        N = L + 1
        self.B, self.L, self.N = B, L, N
        self.in_dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
        if args.ragged:
            lengths = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(7 + rank))
            lengths[0] = L
        else:
            lengths = torch.full((B,), L, dtype=torch.long)
        self.lengths_np = lengths.numpy().copy()
        self.logZ = torch.empty(B, dtype=torch.float32, device=dev)
        self.gdec = torch.zeros((B, N, 2, 2, 2), dtype=torch.float32, device=dev)
        self.gatt = torch.zeros((B, N, N, 2), dtype=torch.float32, device=dev)
        self.n_grad = N * 8 + N * N * 2
        if dry:
            self.launch = lambda: None
            self.count_sum = lambda out: out[:self.n_grad].fill_(1.0)
            return
        from vlgae_amd import _C
        import vlgae_amd.torch_struct as ts
        lib = _C.lib()
        dec, attach, root = synth(B, L, 1000 + rank, dev, torch.float32)
        md32, ma32 = ts.DMV1o.merge(dec, attach, root)                    # merge is the scorer's job (ldndmv.py:209)
        self.md, self.ma = md32.to(self.in_dtype).contiguous(), ma32.to(self.in_dtype).contiguous()
        self.lengths = lengths.to(dev)
        ws_bytes = lib.vlg_workspace_bytes(_C.OP_DMV1O_INSIDE_OUTSIDE, B, N, 0)
        self.ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        dt_code = _C.BF16 if self.in_dtype == torch.bfloat16 else _C.F32
        self.stream = torch.cuda.current_stream(dev)
        sp = _C.ctypes.c_void_p(self.stream.cuda_stream)
        p = [_C.ptr(x) for x in (self.md, self.ma, self.lengths, self.logZ, self.gdec, self.gatt, self.ws)]

        def launch():
            rc = lib.vlg_dmv1o_inside_outside(p[0], p[1], p[2], B, N, dt_code, 0, None, p[3], p[4], p[5], p[6], ws_bytes, sp)
            if rc:
                _C.check(rc, "dmv1o_inside_outside")

        def count_sum(out):   # marginal-loss gradient of position-tied parameters = batch-summed counts, one launch
            rc = lib.vlg_dmv1o_count_sum(p[4], p[5], B, N, _C.ptr(out), sp)
            if rc:
                _C.check(rc, "dmv1o_count_sum")

        self.launch, self.count_sum = launch, count_sum


def run(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    from vlgae_amd import dist as vdist

    dry = os.environ.get("VLGAE_BENCH_DRYRUN") == "1"      # CPU/gloo plumbing test of the launcher path: NO kernels run
    share = os.environ.get("VLGAE_BENCH_SHARE_GPU") == "1"  # debug: every rank on cuda:0 over gloo (never a measurement)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    assert world_env == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world_env}"
    backend = "gloo" if (dry or share) else None            # None -> nccl (= RCCL); chosen without probing the GPU
    rank, local_rank, world = vdist.init_from_env(backend=backend)
    if dry:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
        local_rank = 0 if share else local_rank
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    sync = (lambda: None) if dry else (lambda: torch.cuda.synchronize(dev))

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    if args.workload == "train_step":
        from vlgae_amd.bench import sharded_step as bench_train
        res = bench_train.measure(args, rank, world, dev, dry, barrier)
        if rank == 0:
            print(json.dumps(bench_train.json_line(args, world, res, dry, share)), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    h = Headline(args, rank, dev, dry)
    B, L, N = h.B, h.L, h.N

    def timed_region(reducer, every=1):
        """W untimed warmup steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.
        every: DP launches per collective (1: one all-reduce per launch; 3: the reference's ratio -- it enters the DP three times
        per optimizer step, SURVEY.md section 3.1: lang_feat_max_tree's partition and argmax, the loss's max -- and all-reduces once)."""
        it = [0]

        def step():
            h.launch()
            it[0] += 1
            if reducer is not None and it[0] % every == 0:
                h.count_sum(reducer.buffer)
                reducer.launch()
        for _ in range(args.warmup):
            step()
        if reducer is not None:
            reducer.wait()
        barrier()
        if not dry:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(h.stream)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        if not dry:
            ev1.record(h.stream)
        if reducer is not None:
            reducer.wait()
        barrier()
        elapsed = time.perf_counter() - t0
        gpu_ms = None if dry else ev0.elapsed_time(ev1)
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        return elapsed, gpu_ms

    # Before the first timed region: PREWARM untimed launches of the same kernel (clock ramp, instruction cache, first-touch of the output buffers).
    # The driver's command line has a 3-launch warmup and a 1.4 ms timed region: without this the first region read 4-6 % below the other four
    # (value_spread shows all five either way).  The W warmup steps and the exactly-K timed steps are unchanged.
    if not dry:
        for _ in range(PREWARM):
            h.launch()
        sync()

    comm = {}
    if world > 1:
        comm["rccl_ranks_seen"] = vdist.warm_up(dev)
        comm["backend"] = dist.get_backend()
        n_model = max(h.n_grad, int(args.grad_mb * 1e6 / 4))
        big = vdist.GradAllReducer(n_model, dev)
        # the collective alone: one all-reduce of the model-sized buffer at a time, host-synchronised
        for _ in range(3):
            dist.all_reduce(big.bufs[0])
        barrier()
        t0 = time.perf_counter()
        n_ar = 10
        for _ in range(n_ar):
            dist.all_reduce(big.bufs[0])
        barrier()
        ar_s = (time.perf_counter() - t0) / n_ar
        big.bufs[0].zero_()
        comm.update(allreduce_ms=ar_s * 1e3, allreduce_bytes=n_model * 4,
                    allreduce_busbw_GBs=2.0 * (world - 1) / world * n_model * 4 / ar_s / 1e9)
        elapsed, gpu_ms = timed_region(big)
        spread_s = [elapsed] + [timed_region(big)[0] for _ in range(N_REGIONS - 1)]
        small = vdist.GradAllReducer(h.n_grad, dev)
        elapsed_small, _ = timed_region(small)
        big3 = vdist.GradAllReducer(n_model, dev)
        elapsed_per3, _ = timed_region(big3, every=3)
        # the reduced counts must be the sum over ranks: every rank's attach counts sum to its word count
        total_words = torch.tensor([float(h.lengths_np.sum())], dtype=torch.float64, device=dev)
        dist.all_reduce(total_words)
        if not dry:
            got = float(small.bufs[small.cur ^ 1][N * 8:h.n_grad].sum().item())
            assert abs(got - float(total_words.item())) < 1e-3 * float(total_words.item()), (got, float(total_words.item()))
    else:
        elapsed, gpu_ms = timed_region(None)
        # `value` is the FIRST region (ms_per_step x steps is what was timed); the other regions only quantify the spread of a
        # timed region this short (20 steps = 1.4 ms on the driver's command line)
        spread_s = [elapsed] + [timed_region(None)[0] for _ in range(N_REGIONS - 1)]
        elapsed_small = None

    # ---- multi-GPU: the sharded training step (configs[4]) beside the DP line, on every rank ----
    train_sharded = None
    if world > 1 and not args.no_secondary:
        from vlgae_amd.bench import sharded_step as bench_train
        import copy
        targs = copy.copy(args)
        targs.steps, targs.warmup = min(args.steps, 50), min(args.warmup, 10)
        try:
            train_sharded = bench_train.measure(targs, rank, world, dev, dry, barrier)
            train_sharded.update(steps=targs.steps, warmup=targs.warmup,
                                 what="bench.py --workload train_step on the same ranks (vlgae_amd/bench/sharded_step.py): value = "
                                      "sentences/s of the whole job, synchronous-SGD all-reduce of the model-sized gradient")
        except Exception as e:   # every rank fails or none does (same code path); never costs the headline line
            train_sharded = {"error": repr(e)[:300]}

    # ---- checks outside the timed region: finite, and counts sum to the number of words ----
    if not dry:
        assert bool(torch.isfinite(h.logZ).all()), "non-finite logZ"
        arcs = float(h.gatt.sum().item())
        assert abs(arcs - float(h.lengths_np.sum())) < 1e-3 * h.lengths_np.sum(), (arcs, h.lengths_np.sum())

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    sent_per_s = B * world * args.steps / elapsed
    out = {
        "metric": "sentences/sec, batched inside-outside L=%d B=%d" % (L, B),
        "value": sent_per_s, "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "storage_dtype": args.dtype, "prewarm_launches": 0 if dry else PREWARM,
        "value_spread": dict(zip(("min", "median", "max"), (float(v) for v in np.percentile(
            [B * world * args.steps / e for e in spread_s], [0, 50, 100]))), regions=len(spread_s),
            note=f"{len(spread_s)} timed regions of `steps` steps each (each with its own warmup, barrier and synchronize); "
                 "`value` is the first of them"),
        "data": "synthetic" if not dry else "DRY RUN: no kernels ran (launcher / collective plumbing on CPU, gloo); not a measurement",
        "config": {"workload": "DMV1o inside+outside (Log semiring) -> logZ + expected counts, "
                               f"B={B}/GPU L={L} N={N}, potentials stored {args.dtype}, fp32 charts and arithmetic; "
                               "BASELINE.json configs[1]" + (" sharded x%d, configs[2]" % world if world > 1 else ""),
                   "global_batch": B * world, "seq_len": L, "ragged": bool(args.ragged),
                   "parallelism": (f"dp{world}" if world > 1 else "single") + (" (DEBUG: ranks share one GPU, gloo)" if share else ""),
                   "allreduce_floats": (comm["allreduce_bytes"] // 4 if world > 1 else 0)},
    }
    if train_sharded is not None:
        out["train_step_sharded"] = train_sharded
    if world > 1:
        out["value_dp_grad_only"] = B * world * args.steps / elapsed_small
        out["ms_per_step_dp_grad_only"] = elapsed_small * 1e3 / args.steps
        out["value_allreduce_per_3_launches"] = B * world * args.steps / elapsed_per3
        out["ms_per_step_allreduce_per_3_launches"] = elapsed_per3 * 1e3 / args.steps
        out["comm"] = dict(comm, note="value: EVERY DP launch all-reduces the model-sized flat gradient (--grad-mb, one RCCL call, "
                                      "overlapping the next launch) -- three times the reference's communication per DP entry; "
                                      "value_allreduce_per_3_launches: the same collective once per three DP launches, the reference's "
                                      "ratio (three DP entries per optimizer step, one DDP all-reduce: SURVEY.md 3.1) -- the figure the "
                                      ">= 6x scaling target is to be read on (DESIGN.md section 4); value_dp_grad_only: only the DP's own "
                                      f"{h.n_grad * 4} B of batch-summed counts per launch")
    if dry:
        out["dry_run"] = True
        print(json.dumps(out), flush=True)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    kern_s = gpu_ms * 1e-3 / args.steps                   # HIP events on the launch stream, back-to-back launches
    alg_bytes = algorithmic_bytes(B, N, 2 if args.dtype == "bf16" else 4)
    achieved = alg_bytes / kern_s / 1e9
    exp_ops = exp_class_ops(h.lengths_np)
    exp_peak = N_CU * SIMD_PER_CU * TRANS_LANES_PER_CLK * CLOCK_GHZ * 1e9
    traffic, traffic_src = (None, None)
    if (B, L, args.dtype, bool(args.ragged)) == (256, 40, "bf16", False):
        traffic, traffic_src = pmc_traffic("dmv1o_B256_L40_bf16")
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                       "kernel": "dmv1o_kernel<Log, mode 0 (all charts in LDS), fused inside+outside>",
                       "kernel_us": kern_s * 1e6 if world == 1 else None,
                       "stream_us_per_step": kern_s * 1e6, "algorithmic_bytes_per_launch": alg_bytes,
                       "note": "latency-bound DP: 2(N-1) barrier-separated width steps per sentence, one workgroup "
                               "per sentence; see exp_rate for the bound that binds"
                               + ("" if world == 1 else "; multi-GPU: the stream time per step includes the count-sum launch and waits on the collective")}
    out["exp_rate"] = {"achieved_Gops": exp_ops / kern_s / 1e9, "peak_Gops": exp_peak / 1e9,
                       "frac": exp_ops / kern_s / exp_peak, "ops_per_launch": exp_ops,
                       "note": "exp-class ops (N^3-N inside + 2x outside per sentence) vs v_exp_f32 issue peak "
                               "256 CU x 4 SIMD x 8 lanes/clk x 2.4 GHz"}

    if world == 1 and not args.no_secondary:
        from vlgae_amd.bench import secondary as bench_secondary
        bench_secondary.run_all(out, args, h, dev)

    if world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(B, L, 1000, args.cpu_seconds)
        out["speedup_vs_cpu_baseline"] = sent_per_s / out["cpu_baseline"]["value"]
        if "long_sentence" in out:
            cb = cpu_baseline(B, 80, 1000, min(args.cpu_seconds, 8.0))
            out["long_sentence"]["cpu_baseline"] = cb
            out["long_sentence"]["speedup_vs_cpu_baseline"] = out["long_sentence"]["sentences_per_s"] / cb["value"]
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))
    run(args)


if __name__ == "__main__":
    main()
