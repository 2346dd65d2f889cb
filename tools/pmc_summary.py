"""Folds the passes of tools/prof_kernels.sh into one JSON keyed by the kernels' own (demangled, canonicalised) names:
gpurun_out/TAG_kernels.json.  Per kernel: calls, average duration (kernel-trace stats), every collected counter averaged per launch,
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), per-wave instruction counts, wait / LDS-conflict
fractions, and HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB counters; the gfx950 FETCH_SIZE correction of
/opt/skills/guides/MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

tag, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def canon(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("vlg::", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in name:          # cut the parameter list (the first '(' outside template brackets)
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    name = "".join(out).strip()
    m = re.match(r"_ZN3vlg(\d+)", name)   # a few names come back mangled
    if m:
        k = int(m.group(1))
        name = name[len(m.group(0)):][:k]
    return name


ours = re.compile(r"kernel")
res = collections.defaultdict(dict)
for f in glob.glob(f"gpurun_out/pk_{tag}_stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = canon(r["Name"])
        if "Cijk" in k or k.startswith("at::") or "rocclr" in k or not ours.search(k):
            continue
        res[k].update(calls=int(r["Calls"]), avg_us=float(r["AverageNs"]) / 1e3)
for d in sorted(glob.glob(f"gpurun_out/pk_{tag}_pmc*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = canon(r["Kernel_Name"])
            if k in res:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                res[k].setdefault("counters", {})[c] = sum(v) / len(v)
for k, e in res.items():
    c = e.get("counters", {})
    if c.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        e["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    w = c.get("SQ_WAVES")
    if w:
        e["per_wave"] = {n[9:].lower(): c[n] / w for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if n in c}
    if c.get("SQ_WAVE_CYCLES"):
        e["wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        e["active_inst_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
    if c.get("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes_per_launch"] = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
src = hashlib.sha256()
for f in sorted(glob.glob(os.path.join(ROOT, "vlgae_amd", "csrc", "*"))):
    src.update(open(f, "rb").read())
out = {"how": f"bash tools/prof_kernels.sh {tag} {cmd}: rocprofv3 --kernel-trace --stats, then --pmc passes with --kernel-trace only; "
              "averages per launch; under the profiler clocks run a few % lower than unprofiled",
       "kernel_source_id": src.hexdigest()[:12], "kernels": dict(sorted(res.items()))}
json.dump(out, open(f"gpurun_out/{tag}_kernels.json", "w"), indent=1)
for k, e in sorted(res.items(), key=lambda kv: -kv[1].get("avg_us", 0) * kv[1].get("calls", 0)):
    print(f"{e.get('avg_us', 0):9.1f} us x{e.get('calls', 0):4d}  mfma_busy {e.get('mfma_busy', float('nan')):.3f}  wait {e.get('wait_any_frac', float('nan')):.2f}  "
          f"hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:9.2f} MB  {k[:110]}")
