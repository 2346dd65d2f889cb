"""Register / scratch / LDS figures of every kernel in the built objects, from the gfx950 code-object notes.

    python tools/codeobj.py [pattern]     # kernels whose demangled name contains the pattern

For each translation-unit object in vlgae_amd/_lib the gfx950 offload bundle is extracted (llvm-objdump --offloading, in a
temporary directory) and its AMDGPU metadata note (YAML) parsed: vgpr_count, agpr_count, spills, private segment (scratch)
bytes, static LDS bytes, max workgroup size.  tests/test_codeobj.py asserts on the same records."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj_path):
    """[{name, demangled, vgpr_count, ...}] for one object built by vlgae_amd.build."""
    out = []
    with tempfile.TemporaryDirectory() as td:
        local = os.path.join(td, os.path.basename(obj_path))
        shutil.copy(obj_path, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=td, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL)
        for co in glob.glob(os.path.join(td, "*gfx950*")):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True).stdout
            if "---" not in notes:
                continue
            doc = notes.split("---", 1)[1].split("\n...", 1)[0]
            meta = yaml.safe_load(doc) or {}
            # .text bytes of every kernel: the FUNC symbols of the code object (the kernel descriptor `name.kd` is a separate OBJECT symbol)
            sizes = {}
            for line in subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--symbols", "--wide", co], stdout=subprocess.PIPE, text=True).stdout.splitlines():
                f = line.split()
                if len(f) >= 8 and f[3] == "FUNC":
                    sizes[f[7]] = int(f[2], 0) if not f[2].isdigit() else int(f[2])
            for k in meta.get("amdhsa.kernels", []):
                rec = {key.lstrip("."): val for key, val in k.items() if key != ".args"}
                rec["text_bytes"] = sizes.get(rec.get("name"), 0)
                out.append(rec)
    if out:
        dem = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), stdout=subprocess.PIPE, text=True).stdout.splitlines()
        for k, d in zip(out, dem):
            k["demangled"] = d
    return out


def all_kernels():
    res = []
    for obj in sorted(glob.glob(os.path.join(ROOT, "vlgae_amd", "_lib", "*.o"))):
        for k in kernels_of(obj):
            k["object"] = os.path.basename(obj)
            res.append(k)
    return res


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for k in all_kernels():
        if pat in k.get("demangled", ""):
            print(f'{k["object"]:20s} vgpr {k.get("vgpr_count", -1):3d} agpr {k.get("agpr_count", -1):3d} spill v{k.get("vgpr_spill_count", 0):<3d} s{k.get("sgpr_spill_count", 0):<3d} '
                  f'scratch {k.get("private_segment_fixed_size", 0):4d} text {k.get("text_bytes", 0) / 1024:6.1f}K lds {k.get("group_segment_fixed_size", 0):6d} wg {k.get("max_flat_workgroup_size", 0):4d}  {k.get("demangled", "?")[:120]}')
