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
        if backend == "nccl":
            torch.cuda.set_device(local_rank)   # one process per GPU: RCCL binds the communicator to the current device
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


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
