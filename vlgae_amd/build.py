"""Compile the HIP extension for gfx950 (in-tree, explicit hipcc; no JIT cache).

    python -m vlgae_amd.build            # build if stale
    python -m vlgae_amd.build --force

hipcc cross-compiles without a GPU.  The resulting vlgae_amd/_lib/libvlgae_amd.so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "libvlgae_amd.so")
ARCH = "gfx950"
SOURCES = ("vlg_dp.hip", "vlg_align.hip", "vlg_attn.hip", "vlg_ground.hip", "vlg_decode.hip", "vlg_arc.hip", "vlg_rel.hip", "vlg_feed.cpp", "vlg_capi.cpp")
FLAGS = ("-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-x", "hip")
# per-file extras.  vlg_dp.hip never produces or consumes inf / NaN in arithmetic (the semiring zero is the
# finite -1e12), so fmaxf can be a bare v_max_f32 instead of canonicalise + max.
EXTRA_FLAGS = {"vlg_dp.hip": ("-ffinite-math-only",)}


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X extension cannot be built on this machine")
    return exe


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG, "..", "include", "vlgae_amd.h"),
                                                                 os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    """Build libvlgae_amd.so; returns its path."""
    if not (force or _stale()):
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)

    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(PKG, "..", "include", "vlgae_amd.h"), os.path.abspath(__file__)]

    def compile_one(src):
        obj = os.path.join(LIB_DIR, src + ".o")
        if not force and os.path.exists(obj):   # per-object staleness: its source and every header
            t = os.path.getmtime(obj)
            if all(os.path.getmtime(d) <= t for d in [os.path.join(CSRC, src)] + headers):
                return obj
        cmd = [_hipcc(), f"--offload-arch={ARCH}", *FLAGS, *EXTRA_FLAGS.get(src, ()), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as pool:   # independent translation units
        objs = list(pool.map(compile_one, SOURCES))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
