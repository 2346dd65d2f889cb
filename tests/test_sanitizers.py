"""CPU: the host-side native code under sanitizers.

GPU AddressSanitizer is not available on the pool, so this covers what runs on the host:
  * vlgae_amd/csrc/vlg_feed.cpp   the data feed (k-means, batch builder, the .npy collate thread pool)     ASan+UBSan, TSan
  * tests/emu/emu_dp.cpp           the phase emulator running the DP kernel bodies (vlg_dp_core.h) on
                                   free-running lane threads behind a token barrier                        ASan+UBSan, TSan
  * oracle/vlg_oracle.c            the C oracle (OpenMP over sentences / pairs)                            ASan+UBSan
Each library is rebuilt with -fsanitize=... into a git-ignored _build directory and the EXISTING pytest files of that component
are run in a child interpreter with the sanitizer runtime preloaded and the library path overridden (VLGAE_AMD_LIB / VLG_EMU_SO /
VLG_ORACLE_SO); the child must pass and print no sanitizer report.  (TSan is not applied to the oracle: libgomp is not
TSan-instrumented and reports its own barriers.)"""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BUILD = os.path.join(ROOT, "tests", "emu", "_build")
CSRC = os.path.join(ROOT, "vlgae_amd", "csrc")
REPORTS = ("AddressSanitizer", "ThreadSanitizer", "runtime error:", "LeakSanitizer")
SAN = {"asan": ("-fsanitize=address,undefined", "libasan.so"), "tsan": ("-fsanitize=thread", "libtsan.so")}


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
    if not os.path.isabs(path) or not os.path.exists(path):
        pytest.skip(f"{name} is not installed with this gcc")
    return path


def _build(kind, name, compiler, sources, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, f"lib{name}_{kind}.so")
    deps = list(sources) + [os.path.join(CSRC, "vlg_dp_core.h"), os.path.join(CSRC, "vlg_common.h"),
                            os.path.join(ROOT, "oracle", "vlg_oracle_impl.h"), os.path.abspath(__file__)]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        cmd = [compiler, "-O1", "-g", "-fPIC", "-shared", "-fno-omit-frame-pointer", SAN[kind][0], *extra, *sources, "-o", out]
        subprocess.run(cmd, check=True)
    return out


def _child(kind, env_lib, tests, k=None, timeout=900):
    env = dict(os.environ, LD_PRELOAD=_runtime(SAN[kind][1]), OMP_NUM_THREADS="4", **env_lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:second_deadlock_stack=1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "not gpu", "-p", "no:cacheprovider", *tests]
    if k:
        cmd += ["-k", k]
    proc = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = proc.stdout[-4000:]
    assert proc.returncode == 0, tail
    assert " passed" in proc.stdout and not any(r in proc.stdout for r in REPORTS), tail
    return proc.stdout


FEED_SRC = [os.path.join(CSRC, "vlg_feed.cpp"), os.path.join(CSRC, "vlg_capi.cpp")]
FEED_FLAGS = ("-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include")   # host code only: no HIP call is linked


@pytest.mark.timeout(900)
@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_feed_under_sanitizers(kind):
    """The whole of tests/test_feed.py (sampler epochs, k-means vs oracle, threaded .npy collate incl. ragged / sampled region
    lists) against a sanitized build of the feed translation unit loaded through the product's own binding (VLGAE_AMD_LIB)."""
    lib = _build(kind, "vlg_feed", "g++", FEED_SRC, FEED_FLAGS)
    out = _child(kind, {"VLGAE_AMD_LIB": lib}, ["tests/test_feed.py"])
    assert "passed" in out


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("kind", ["asan", "tsan"])
def test_dp_kernel_bodies_in_the_emulator_under_sanitizers(kind):
    """tests/test_emu_kernel_bodies.py on the small fixtures: the kernels' per-thread bodies on nt host threads, every thread order
    -- out-of-bounds chart accesses (ASan: the arenas are heap blocks), signed overflow / bad shifts in the index math (UBSan), and
    unsynchronised accesses between lane threads across the token barrier (TSan)."""
    lib = _build(kind, "vlg_emu", "g++", [os.path.join(ROOT, "tests", "emu", "emu_dp.cpp")], ("-std=c++17", "-pthread"))
    out = _child(kind, {"VLG_EMU_SO": lib}, ["tests/test_emu_kernel_bodies.py"], k="L10 or L7 or L1_ or L2_ or N5 or N6 or N2 or ties or decode",
                 timeout=1400)
    assert "passed" in out


@pytest.mark.timeout(1500)
def test_c_oracle_under_asan_ubsan():
    """tests/test_oracle_golden.py (every restated function against the reference-made fixtures) on an ASan + UBSan build of
    oracle/vlg_oracle.c."""
    lib = _build("asan", "vlg_oracle", "gcc", [os.path.join(ROOT, "oracle", "vlg_oracle.c")], ("-std=c11", "-fopenmp", "-fno-fast-math", "-lm"))
    out = _child("asan", {"VLG_ORACLE_SO": lib}, ["tests/test_oracle_golden.py"], timeout=1400)
    assert "passed" in out
