"""CPU, world_size 2, gloo: the multi-GPU path shards sentences across ranks and combines the
marginal-loss gradient with ONE all-reduce.  The per-rank compute here is the oracle (the HIP path needs
a GPU); what is under test is vlgae_amd.dist -- sharding, the double-buffered asynchronous reducer and
its equivalence with the unsharded batch."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import oracle
    from vlgae_amd import dist as vdist
    oracle.set_threads(1)
    r, lr, w = vdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"

    rng = np.random.default_rng(0)                 # every rank draws the same global batch
    B, L = 10, 9                                   # 10 over 2 ranks = 5 + 5; also try an uneven split below
    dec = rng.standard_normal((B, L, 2, 2, 2)).astype(np.float32)
    attach = rng.standard_normal((B, L, L, 2)).astype(np.float32)
    root = rng.standard_normal((B, L)).astype(np.float32)
    lengths = rng.integers(1, L + 1, size=B)
    md, ma = oracle.dmv1o_merge(dec, attach, root)
    N = L + 1
    n_grad = N * 8 + N * N * 2

    results = {}
    for n_items in (B, B - 3):                     # 7 items -> 4 + 3
        s, e = vdist.shard_bounds(n_items, rank, world)
        md_s, ma_s, ln_s = vdist.shard_batch([torch.from_numpy(md[:n_items]), torch.from_numpy(ma[:n_items]),
                                              torch.from_numpy(lengths[:n_items])], rank, world)
        assert md_s.shape[0] == e - s
        red = vdist.GradAllReducer(n_grad, torch.device("cpu"))
        steps = []
        for step in range(3):                      # several steps: exercises the double buffering
            scale = float(step + 1)
            lz, gd, ga = oracle.dmv1o(md_s.numpy(), ma_s.numpy(), ln_s.numpy(), "log", np.float32,
                                      glogZ=np.full(e - s, scale, np.float32))
            buf = red.buffer
            buf[:N * 8] = torch.from_numpy(gd.reshape(e - s, -1).sum(0))
            buf[N * 8:] = torch.from_numpy(ga.reshape(e - s, -1).sum(0))
            reduced = red.launch()                 # asynchronous; flips to the other buffer
            if step == 1:                          # two collectives in flight at once (steps 0 and 1)
                red.wait()
                steps.append(prev.clone())
                steps.append(reduced.clone())
            elif step == 2:
                red.wait()
                steps.append(reduced.clone())
            prev = reduced
        _, gd_all, ga_all = oracle.dmv1o(md[:n_items], ma[:n_items], lengths[:n_items], "log", np.float32)
        full = np.concatenate([gd_all.reshape(n_items, -1).sum(0), ga_all.reshape(n_items, -1).sum(0)])
        errs = [float(np.abs(steps[k].numpy() - (k + 1) * full).max()) for k in range(3)]
        results[n_items] = errs
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.array([results[B], results[B - 3]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_gradient_allreduce_matches_full_batch(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        errs = np.load(tmp_path / f"rank{rank}.npy")
        assert errs.shape == (2, 3) and errs.max() <= 2e-4, errs     # sums of ~10 fp32 expected-count tensors


def _run_bench(cmd, extra_env):
    import json
    import subprocess
    env = dict(os.environ, VLGAE_BENCH_DRYRUN="1", OMP_NUM_THREADS="1", **extra_env)
    env.pop("WORLD_SIZE", None)
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert proc.returncode == 0, proc.stderr[-3000:]
    # gloo's own C++ chatter ("[Gloo] Rank 0 is connected to ...") goes to stdout too, unsynchronised between the ranks
    # (fragments of it can land on lines of their own): what must hold is exactly ONE JSON line, from rank 0
    lines = [ln for ln in proc.stdout.splitlines() if ln.lstrip().startswith("{")]
    assert len(lines) == 1 and proc.stdout.count('"metric"') == 1, proc.stdout
    return json.loads(lines[0])


def _check_dry_line(out, world):
    assert out["dry_run"] is True and "DRY RUN" in out["data"]      # can never be mistaken for a measurement
    assert out["n_gpus"] == world and out["scaling"] == "weak" and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 256 * world and out["config"]["parallelism"] == f"dp{world}"
    assert out["comm"]["rccl_ranks_seen"] == world and out["comm"]["backend"] == "gloo"
    assert out["config"]["allreduce_floats"] == 125000 and out["comm"]["allreduce_bytes"] == 500000
    assert out["comm"]["allreduce_ms"] > 0 and out["value"] > 0 and out["value_dp_grad_only"] > 0
    assert out["value_allreduce_per_3_launches"] > 0 and "per three DP launches" in out["comm"]["note"]


@pytest.mark.timeout(300)
def test_bench_self_launches_ranks_and_relays_one_json_line():
    """`python bench.py --gpus 2` with no torchrun around it: the parent (which never touches a GPU) starts the rank
    processes, rank 0's JSON line comes back on stdout.  VLGAE_BENCH_DRYRUN=1 swaps the kernels for no-ops and RCCL for
    gloo so the launcher / process-group / reducer path runs on CPU."""
    out = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--grad-mb", "0.5",
                      "--cpu-seconds", "0"], {})
    _check_dry_line(out, 2)


@pytest.mark.timeout(300)
def test_bench_under_torchrun_as_the_driver_launches_it():
    out = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                      "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2",
                      "--steps", "3", "--warmup", "1", "--grad-mb", "0.5", "--cpu-seconds", "0"], {})
    _check_dry_line(out, 2)


def test_bench_parent_does_not_import_torch_before_spawning():
    """The self-launching parent must not initialise the GPU: it decides from argv/env alone, before `import torch`."""
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    top_imports = {a.name.split(".")[0] for n in tree.body if isinstance(n, ast.Import) for a in n.names}
    top_imports |= {n.module.split(".")[0] for n in tree.body if isinstance(n, ast.ImportFrom)}
    assert "torch" not in top_imports and "vlgae_amd" not in top_imports and "numpy" not in top_imports


@pytest.mark.timeout(300)
def test_bench_train_step_workload_sharded_dry_run():
    """`bench.py --workload train_step --gpus 2`: the sharded training step (configs[4]) -- bucketed synchronous-SGD
    all-reduce of the flat gradient, tail bucket started from inside the backward pass, sum-over-ranks check."""
    out = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--grad-mb", "12",
                      "--cpu-seconds", "0", "--workload", "train_step"], {})
    assert out["dry_run"] is True and "DRY RUN" in out["data"]
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and "configs[4]" in out["config"]["workload"]
    assert out["config"]["global_batch"] == 512 and out["config"]["parallelism"] == "dp2"
    comm = out["comm"]
    assert comm["rccl_ranks_seen"] == 2 and comm["backend"] == "gloo" and comm["buckets"] == 4
    # the whole shipped model: 6.48 M trainable floats (text encoder, three visual-encoder MLPs, ... -- no filler), whatever --grad-mb says
    assert comm["allreduce_bytes"] == out["config"]["allreduce_floats"] * 4 and 6_300_000 <= out["config"]["allreduce_floats"] <= 7_500_000
    assert "no filler" in comm["payload"]
    # four buckets in readiness order: the arc encoder's (w1 | w2 | b), the parser's feed-forwards, the language-side encoders / LayerNorm /
    # pre-matching projection, then the check slot + the text and visual encoders (+ the zeros of the visual-encoder MLPs the object-only
    # layout does not reach); together they tile the buffer
    b0, b1, b2, b3 = comm["bucket_bounds"]
    assert b0 == [0, 128 ** 3 + 128 * 128 + 128] and b1[0] == b0[1] and b2[0] == b1[1] and b3[0] == b2[1] and b3[1] == out["config"]["allreduce_floats"]
    assert comm["bucket_contents"][0] == ["w1", "w2", "b"] and "w_vis" in comm["bucket_contents"][2] and comm["bucket_contents"][3] == ["w_text", "w_venc", "b_venc"]
    assert out["real_gradient_floats"] == out["config"]["allreduce_floats"] - 1 - 2 * 256 * (4096 + 1)
    chk = comm["mean_over_ranks_check"]
    assert chk["slot"] == chk["expected"] == 256 * 40        # DDP averages: the mean of the two ranks' word counts
    assert out["value"] > 0 and out["step_ms"] > 0 and out["compute_ms"] > 0 and comm["allreduce_ms"] > 0


@pytest.mark.timeout(300)
def test_bench_dp_line_carries_the_sharded_train_step():
    """The default (DP) workload on N > 1 ranks also runs the sharded training step, so the driver's scaling run records
    both without being asked."""
    out = _run_bench([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--grad-mb", "10",
                      "--cpu-seconds", "0"], {})
    ts = out["train_step_sharded"]
    assert "error" not in ts, ts
    assert ts["comm"]["rccl_ranks_seen"] == 2 and ts["comm"]["buckets"] == 4 and ts["value"] > 0


def _bucket_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from vlgae_amd import dist as vdist
    vdist.init_from_env(backend="gloo")
    ok = True
    for numel, nb, head in ((1000, 2, 10), (1000, 3, 700), (7, 4, 3), (5, 1, 5), (64, 2, 64)):
        red = vdist.BucketedGradReducer(numel, torch.device("cpu"), n_buckets=nb, head=head, average=(numel != 7))
        # the buckets tile [0, numel) exactly once; the head sits in the bucket that is launched last
        cover = sorted(red.bounds)
        ok &= cover[0][0] == 0 and cover[-1][1] == numel and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
        ok &= red.bounds[-1][0] == 0 and red.bounds[-1][1] >= head
        for step in range(2):
            red.wait()
            red.flat.copy_(torch.arange(numel, dtype=torch.float32) * (rank + 1) * (step + 1))
            for i in range(red.n_buckets):
                red.launch(i)
        red.wait()
        # DDP semantics by default: the MEAN over ranks (one case asks for the plain sum)
        want = torch.arange(numel, dtype=torch.float32) * 2 * sum(r + 1 for r in range(world)) / (1 if numel == 7 else world)
        ok &= bool(torch.allclose(red.flat, want, rtol=1e-6, atol=0))
    # explicit bounds in launch order (what vlgae_amd/bench/sharded_step.py passes): must tile the buffer
    red = vdist.BucketedGradReducer(10, torch.device("cpu"), bounds=[(0, 4), (4, 6), (6, 10)])
    ok &= red.n_buckets == 3 and red.bounds == [(0, 4), (4, 6), (6, 10)]
    try:
        vdist.BucketedGradReducer(10, torch.device("cpu"), bounds=[(0, 4), (5, 10)])
        ok = False
    except ValueError:
        pass
    np.save(os.path.join(out_dir, f"bucket{rank}.npy"), np.array([int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_reducer_tiles_the_buffer_and_averages_over_ranks(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_bucket_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        assert int(np.load(tmp_path / f"bucket{rank}.npy")[0]) == 1
