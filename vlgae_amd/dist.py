"""Multi-GPU: one process per GPU, sentences sharded across ranks, ONE RCCL all-reduce of the
marginal-loss gradient per step.

The DP treats sentences as independent (batch is a pure map dimension in dmv.py:19-66 / deptree.py:25-76),
so ranks never exchange chart data.  The only collective is the data-parallel gradient sum the reference
gets implicitly from Lightning DDP (config/trainer/train.yaml:27-29); here it is one flat buffer and one
`all_reduce(SUM)` on the `nccl` backend (= RCCL over xGMI on ROCm), issued asynchronously so that it
overlaps the next step's kernels.  On CPU (tests) the same code runs on `gloo`.
"""
import os

# The host driver only supports dmabuf IPC; HSA reads this flag ONCE, when the runtime initialises (the first HIP call
# of the process, torch.cuda.is_available() included), so it has to be in the environment before anything below can
# touch the GPU.  Launchers that import this module late should export it themselves (bench.py does, at its top).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from torchrun's environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    Returns (rank, local_rank, world_size).  world_size == 1 needs no process group.

    backend: "nccl" (= RCCL over xGMI, the product path) unless the caller or VLGAE_DIST_BACKEND says otherwise
    ("gloo": CPU tests).  It is chosen from arguments / environment only -- never by probing the GPU, which would
    initialise HSA before the process group exists."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        backend = backend or os.environ.get("VLGAE_DIST_BACKEND", "nccl")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)   # one process per GPU: RCCL binds the communicator to the current device
            # device_id makes the communicator eager: it is built here, not inside the first collective of a timed section
            kw["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        warm_up(torch.device("cuda", local_rank) if backend == "nccl" else torch.device("cpu"))
    return rank, local_rank, world


def warm_up(device):
    """One small all-reduce + barrier: communicator set-up, xGMI ring discovery and the first-call kernel loads of RCCL
    happen here, before anything is timed.  Returns the number of ranks the collective saw (must equal world_size)."""
    ones = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce(ones)
    seen = int(round(float(ones.item())))
    if seen != dist.get_world_size():
        raise RuntimeError(f"warm-up all-reduce saw {seen} ranks, world size is {dist.get_world_size()}")
    return seen


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced shard [start, end) of n_items for `rank` (first n_items % world ranks get +1)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batch(tensors, rank, world):
    """Slice every tensor of a batch along dim 0 to this rank's shard."""
    n = tensors[0].shape[0]
    s, e = shard_bounds(n, rank, world)
    return [t[s:e] for t in tensors]


class GradAllReducer:
    """Double-buffered flat gradient + one asynchronous all-reduce(SUM) per step.

    pack(parts) copies/accumulates gradient pieces into the current flat buffer; launch() starts the
    collective and flips buffers; wait() blocks the *stream* (not the host) on the previous one."""

    def __init__(self, numel, device, dtype=torch.float32, average=False):
        self.bufs = [torch.zeros(numel, dtype=dtype, device=device) for _ in range(2)]
        self.handles = [None, None]
        self.cur = 0
        self.average = average
        self.world = dist.get_world_size() if dist.is_initialized() else 1

    @property
    def buffer(self):
        return self.bufs[self.cur]

    def launch(self):
        """All-reduce the current buffer asynchronously; returns the buffer being reduced."""
        buf = self.bufs[self.cur]
        if self.world > 1:
            self.handles[self.cur] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        self.cur ^= 1
        self.wait(self.cur)   # the buffer we are about to reuse must have finished its previous reduce
        return buf

    def wait(self, which=None):
        for i in ([which] if which is not None else [0, 1]):
            h = self.handles[i]
            if h is not None:
                h.wait()
                self.handles[i] = None
                if self.average:
                    self.bufs[i].div_(self.world)


class BucketedGradReducer:
    """The training step's flat gradient reduced in contiguous pieces with synchronous-SGD semantics (what Lightning DDP does for
    the reference, config/trainer/train.yaml:27-29): bucket i's all-reduce is started as soon as its gradients exist (`launch(i)`,
    asynchronous, on RCCL's own stream, ordered after the kernels already enqueued on the current stream), and `wait()` -- called
    before the optimizer / the next step -- makes the current stream (not the host) wait for all of them.  Buckets launched early
    overlap the rest of the backward pass.  average=True (default): the reduced buffer holds the MEAN over ranks, as DDP's does
    (`ReduceOp.AVG` on RCCL; sum, then one division at `wait()` on backends without it)."""

    def __init__(self, numel, device, n_buckets=2, dtype=torch.float32, head=0, bounds=None, average=True):
        """bounds: explicit [(lo, hi), ...] in LAUNCH order (a partition of [0, numel)).  Otherwise -- head: number of leading
        elements that must sit in the LAST-launched bucket's view; the rest of the buffer is dealt evenly."""
        self.flat = torch.zeros(numel, dtype=dtype, device=device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.average = bool(average)
        self._native_avg = self.average and dist.is_initialized() and dist.get_backend() == "nccl"
        if bounds is not None:
            cover = sorted(bounds)
            if cover[0][0] != 0 or cover[-1][1] != numel or any(a[1] != b[0] for a, b in zip(cover, cover[1:])):
                raise ValueError(f"bucket bounds {bounds} do not tile [0, {numel})")
            self.bounds = [tuple(b) for b in bounds]
        else:
            n_buckets = max(1, min(n_buckets, numel))
            # last bucket = [0, cut_last) holds the head; earlier buckets split the tail evenly
            per = -(-numel // n_buckets)
            last = max(per, head)
            bounds = [(0, min(last, numel))]
            rest = numel - bounds[0][1]
            k = n_buckets - 1
            lo = bounds[0][1]
            for i in range(k):
                hi = lo + (rest // k) + (1 if i < rest % k else 0)
                if hi > lo:
                    bounds.append((lo, hi))
                lo = hi
            # launch order: early buckets first, the head-carrying bucket last
            self.bounds = bounds[1:] + bounds[:1]
        self.views = [self.flat[a:b] for a, b in self.bounds]
        self.handles = []
        self._launched = []

    @property
    def n_buckets(self):
        return len(self.views)

    @property
    def head_view(self):
        return self.views[-1]

    def launch(self, i):
        if self.world > 1:
            op = dist.ReduceOp.AVG if self._native_avg else dist.ReduceOp.SUM
            self.handles.append(dist.all_reduce(self.views[i], op=op, async_op=True))
            self._launched.append(i)

    def wait(self):
        for h in self.handles:
            h.wait()
        if self.average and not self._native_avg:
            for i in self._launched:
                self.views[i].div_(self.world)
        self.handles = []
        self._launched = []
