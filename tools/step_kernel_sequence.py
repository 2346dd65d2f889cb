"""The kernel sequence of ONE training step out of a rocprofv3 kernel trace (tools/prof_train_step.sh):
    python tools/step_kernel_sequence.py gpurun_out/prof_TAG_ts [--all]
prints duration (us) and name of every launch of the last graph replay, the totals, and how much of the step is launches shorter
than 8 us (the fold-able tail)."""
import csv, glob, re, sys

d = sys.argv[1]
import os
f = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)   # the newest run in the directory
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
ends = [i for i, n in enumerate(names) if "rng_advance_kernel" in n]             # round 5: the step's LAST launch (the dropout generator's step counter)
if len(ends) >= 2:
    seq = rows[ends[-2] + 1:ends[-1] + 1]
else:                                                                            # round 3's chain (no encoders in front): once-per-step marker + walk back
    marks = [i for i, n in enumerate(names) if "attn_fuse_mfma_kernel" in n]
    a, b = marks[-2], marks[-1]
    back = 0
    while back < 40 and "attn_fuse_bwd" not in names[a - back - 1] and "gemm_reduce" not in names[a - back - 1]:
        back += 1                                                                # the step starts behind the previous step's last adjoint
    seq = rows[a - back:b - back]


def short(n):
    n = re.sub(r"void |at::native::|\(anonymous namespace\)::|vlg::", "", n)
    n = re.sub(r"vectorized_elementwise_kernel<\d+, ", "vec<", n)
    n = re.sub(r"elementwise_kernel_manual_unroll<\d+, \d+, gpu_kernel_impl_nocast<", "ew<", n)
    return n[:110]


dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seq]
if "--all" in sys.argv:
    for t, r in zip(dur, seq):
        print(f"{t:7.1f}  {short(r['Kernel_Name'])}")
wall = (int(seq[-1]["End_Timestamp"]) - int(seq[0]["Start_Timestamp"])) / 1e3
small = [t for t in dur if t < 8]
print(f"{len(seq)} launches, wall {wall:.0f} us, kernel time {sum(dur):.0f} us; {len(small)} launches < 8 us = {sum(small):.0f} us")
