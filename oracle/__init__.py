"""CPU oracle for the VLGAE structured-DP hot path -- TEST INFRASTRUCTURE, not product code.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this
package.  The product (`vlgae_amd`) never does, and raises if its HIP library is missing.
"""
from .cpu_oracle import (  # noqa: F401
    NEGINF, attn_fuse, bilinear_align, build, deptree, dmv1o, dmv1o_merge, enumerate_dmv1o,
    enumerate_deptree, max_threads, set_threads,
)
