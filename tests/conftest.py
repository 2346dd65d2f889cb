import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def golden_ids(prefix):
    return [os.path.basename(f)[:-4] for f in golden_files(prefix)]


def load(path):
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def arcenc_w1(g):
    """w1 of an arc-encoder fixture: stored whole, or as rank factors for the large case (make_golden.arcenc_cases)."""
    if "w1" in g:
        return g["w1"]
    return np.einsum("xr,hr,yr->xhy", g["w1_u"].astype(np.float64), g["w1_v"].astype(np.float64),
                     g["w1_z"].astype(np.float64)).astype(np.float32)


def arcenc_check_w1_grad(got, g, tol):
    if "g_w1" in g:
        ref = g["g_w1"]
    else:
        got, ref = got[::5, ::7, ::3], g["g_w1_sample"]
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), "g_w1"


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
