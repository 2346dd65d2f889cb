"""Drop-in replacement for `src.model.torch_struct` on the path VLGAE exercises (see distributions.py)."""
from .distributions import DMV1o, DMV1oRules, DependencyCRF, StructDistribution
from .dmv import DEC_NUM, DIR_NUM, GO, HASCHILD, LEFT, NOCHILD, RIGHT, STOP, VAL_NUM
from .semirings import NEGINF, LogSemiring, MaxSemiring

version = "0.4"  # API level of the vendored pytorch-struct the reference ships (torch_struct/__init__.py:19)

__all__ = ["DMV1o", "DMV1oRules", "DependencyCRF", "StructDistribution", "LogSemiring", "MaxSemiring", "NEGINF", "NOCHILD",
           "HASCHILD", "LEFT", "RIGHT", "GO", "STOP", "DIR_NUM", "VAL_NUM", "DEC_NUM"]


# Backward on the calling thread, by default (round 4).  torch hands the backward pass of a GPU graph to a per-device
# engine thread; that hand-off is a condition-variable wake-up of 50-120 us per `backward()` / `autograd.grad()` on the hosts measured --
# more than the fused inside+outside kernel takes -- so an UNCHANGED trainer that swaps this package in for `src.model.torch_struct`
# would run the DP at half its speed (3.07 M vs 1.24-1.6 M sentences/s through the API).  The reference runs one process per GPU
# (Lightning DDP, config/trainer/train.yaml:27-29), which has no use for that thread, so importing the drop-in package selects
# `torch.autograd.set_multithreading_enabled(False)` for the process.  Opt out with VLGAE_AMD_AUTOGRAD_THREAD=engine (then the
# one-time warning of functional._note_backward_thread applies); `vlgae_amd.configure_autograd()` does the same explicitly.
import os as _os

if _os.environ.get("VLGAE_AMD_AUTOGRAD_THREAD", "caller").lower() != "engine":
    import logging as _logging
    import torch as _torch
    if _torch.autograd.is_multithreading_enabled():
        # a process-wide setting changed by an import: say so once, with the way out -- multi-device-per-process users
        # (DataParallel, model parallel) want torch's per-device engine threads back
        _logging.getLogger("vlgae_amd").info(
            "vlgae_amd.torch_struct: autograd backward now runs on the calling thread for this process "
            "(torch.autograd.set_multithreading_enabled(False)); set VLGAE_AMD_AUTOGRAD_THREAD=engine before the import to keep torch's "
            "per-device engine threads")
        _torch.autograd.set_multithreading_enabled(False)
