"""Diagnostic: which torch (at::native) kernels run inside the training step, by total time.  Usage: torch_glue.py trace.csv"""
import csv, sys, re
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "at::native" not in n:
        continue
    m = re.findall(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", n)
    key = " / ".join(dict.fromkeys(m[:4]))
    agg[key][0] += 1
    agg[key][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:7.2f} ms  {c:5d} x {t/c/1e3:7.1f} us  {k[:150]}")
