"""Import shim used ONLY by make_golden.py (runs in the build container, never on the GPU box).

The reference (/root/reference) is a Hydra / Lightning / fastNLP program; none of those
packages are installed here.  The structured-DP half (src/model/torch_struct) imports
cleanly on its own.  The alignment half (src/model/joint.py) needs its third-party
imports satisfied; we register inert stub modules for them so that the reference's own
`DependencyBoxRel.gather_logit_simple` can be executed to produce golden vectors.

Nothing here is product code and nothing here is copied from the reference.
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest import mock

REFERENCE_ROOT = "/root/reference"

_STUB_PREFIXES = (
    "hydra", "omegaconf", "pytorch_lightning", "easydict", "colorama", "fastNLP",
    "nltk", "wandb", "torchmetrics", "torchvision", "seaborn", "spacy", "nni",
)


class _StubMeta(type):
    """Metaclass: unknown class attributes (e.g. OmegaConf.register_new_resolver) are mocks."""

    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return mock.MagicMock(name=f"{cls.__name__}.{name}")


class _StubModule(types.ModuleType):
    """Module whose unknown attributes are MagicMocks (classes where subclassing needs it)."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        # things that get subclassed / used as decorators must be real classes
        val = _StubMeta(name, (), {"__init__": lambda self, *a, **k: None})
        if name and name[0].islower():
            val = mock.MagicMock(name=f"{self.__name__}.{name}")
        setattr(self, name, val)
        return val


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_PREFIXES:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def import_torch_struct():
    """The DP half: only relative imports + torch."""
    p = REFERENCE_ROOT + "/src/model"
    if p not in sys.path:
        sys.path.insert(0, p)
    import torch_struct  # noqa
    return torch_struct


def import_joint():
    """The alignment half, through stubbed third-party packages."""
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    import easydict
    easydict.EasyDict = dict
    import hydra._internal.utils as hiu
    hiu.is_under_debugger = lambda: False
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import src  # noqa
    from src.model import joint
    return src, joint
