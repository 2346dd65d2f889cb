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


def gdecode_check_lists(got_factor, got_img, g, logit_after):
    """Compare the two result lists of decode_grounding_on_factor with the fixture's (joint.py:598-629).  Candidate k of a
    row is compared only where its value is not tied with a neighbour in the sorted row: torch.argsort leaves the order
    of equal values open, and the edited rows hold many equal fills (-1e10, -1e20)."""
    import json
    want_factor = json.loads(str(g["txt_to_factor"]))
    want_img = json.loads(str(g["txt_to_img"]))
    assert [[int(t) for t in row] for row in got_img] == want_img
    tmask = g["tmask"]
    assert len(got_factor) == len(want_factor)
    n_cmp = 0
    for b in range(len(want_factor)):
        rows = [q for q in range(tmask.shape[1]) if tmask[b, q]]
        assert len(got_factor[b]) == len(want_factor[b]) == len(rows)
        for r, q in enumerate(rows):
            v = np.sort(logit_after[b, q])[::-1]
            assert len(got_factor[b][r]) == len(want_factor[b][r]) == min(5, len(v))
            for k in range(min(5, len(v))):
                tied = (k + 1 < len(v) and v[k] == v[k + 1]) or (k > 0 and v[k] == v[k - 1])
                if tied:
                    continue
                got = json.loads(json.dumps(got_factor[b][r][k]))     # tuples -> lists, like the fixture
                assert got == want_factor[b][r][k], (b, q, k)
                n_cmp += 1
    assert n_cmp > 0


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle
