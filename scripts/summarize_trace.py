"""Summarise a rocprofv3 --kernel-trace CSV into a per-kernel table for the DECODE phase only
(launches after the first step_prep_kernel), with inter-kernel gaps.  Usage:
    python scripts/summarize_trace.py gpurun_out/prof_x/x_kernel_trace.csv > profiles/r01_decode_kernels.md
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "step_prep_kernel" in r["Kernel_Name"])
dec = rows[first:]
nsteps = sum(1 for r in dec if "step_prep_kernel" in r["Kernel_Name"])


def short(n):
    n = n.replace("void ", "")
    return n[: n.index("(")] if "(" in n else n[:60]


agg = defaultdict(lambda: [0, 0, 10**12, 0])
gap_total = 0
for i, r in enumerate(dec):
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg[short(r["Kernel_Name"])]
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
    if i:
        g = int(r["Start_Timestamp"]) - int(dec[i - 1]["End_Timestamp"])
        if 0 < g < 200000:
            gap_total += g
span = int(dec[-1]["End_Timestamp"]) - int(dec[0]["Start_Timestamp"])
busy = sum(a[1] for a in agg.values())
print(f"# decode phase: {nsteps} steps, {len(dec)} launches, span {span/1e6:.2f} ms = {span/1e3/nsteps:.1f} us/step; "
      f"kernel time {busy/1e3/nsteps:.1f} us/step, gaps {gap_total/1e3/nsteps:.1f} us/step\n")
print("| kernel | launches/step | avg us | min us | max us | us/step | % of kernel time |")
print("|---|---|---|---|---|---|---|")
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| `{k}` | {a[0]/nsteps:.1f} | {a[1]/a[0]/1e3:.2f} | {a[2]/1e3:.2f} | {a[3]/1e3:.2f} | {a[1]/nsteps/1e3:.1f} | {100*a[1]/busy:.1f} |")
