"""Drop-in replacement for `src.model.torch_struct` on the path VLGAE exercises (see distributions.py)."""
from .distributions import DMV1o, DMV1oRules, DependencyCRF, StructDistribution
from .dmv import DEC_NUM, DIR_NUM, GO, HASCHILD, LEFT, NOCHILD, RIGHT, STOP, VAL_NUM
from .semirings import NEGINF, LogSemiring, MaxSemiring

version = "0.4"  # API level of the vendored pytorch-struct the reference ships (torch_struct/__init__.py:19)

__all__ = ["DMV1o", "DMV1oRules", "DependencyCRF", "StructDistribution", "LogSemiring", "MaxSemiring", "NEGINF", "NOCHILD",
           "HASCHILD", "LEFT", "RIGHT", "GO", "STOP", "DIR_NUM", "VAL_NUM", "DEC_NUM"]
