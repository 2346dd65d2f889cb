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
FLAGS = ("-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-x", "hip")
# The DP kernels never produce or consume inf / NaN in arithmetic (the semiring zero is the finite -1e12), so fmaxf
# can be a bare v_max_f32 instead of canonicalise + max.
DP_FLAGS = ("-ffinite-math-only",)
# Translation units: (source, object name, extra flags).  The structured-DP kernel templates (vlg_dp_kernels.h) are
# instantiated by vlg_dp_inst.hip, compiled once per (family: DMV1o merged / DMV1o rules / DepTree) x (semiring) x
# (input type) -- 12 objects of 8 kernels each instead of one 100-kernel object that took 5.5 minutes on one core.
DP_INST = tuple(("vlg_dp_inst.hip", f"vlg_dp_inst_{f}{s}{i}", DP_FLAGS + (f"-DVLG_INST_FAMILY={f}", f"-DVLG_INST_SR={s}", f"-DVLG_INST_IN={i}"))
                for f in (0, 1, 2) for s in (0, 1) for i in (0, 1))
UNITS = DP_INST + (("vlg_dp.hip", "vlg_dp", DP_FLAGS), ("vlg_dp_pair.hip", "vlg_dp_pair", DP_FLAGS)) + tuple(
    (src, os.path.splitext(src)[0], ()) for src in
    ("vlg_align.hip", "vlg_attn.hip", "vlg_ground.hip", "vlg_decode.hip", "vlg_arc.hip", "vlg_rel.hip", "vlg_gemm.hip", "vlg_langfeat.hip", "vlg_ff.hip", "vlg_ffgemm.hip", "vlg_encoders.hip", "vlg_scorer.hip", "vlg_feed.cpp", "vlg_capi.cpp"))
SOURCES = tuple(sorted({u[0] for u in UNITS}))
JOBS = max(1, min(len(UNITS), int(os.environ.get("VLGAE_BUILD_JOBS", os.cpu_count() or 4))))


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

    def compile_one(unit):
        src, name, extra = unit
        obj = os.path.join(LIB_DIR, name + ".o")
        if not force and os.path.exists(obj):   # per-object staleness: its source and every header
            t = os.path.getmtime(obj)
            if all(os.path.getmtime(d) <= t for d in [os.path.join(CSRC, src)] + headers):
                return obj
        cmd = [_hipcc(), f"--offload-arch={ARCH}", *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=JOBS) as pool:   # independent translation units
        objs = list(pool.map(compile_one, UNITS))
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB_PATH + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
