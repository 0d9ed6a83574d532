"""Per-kernel totals of a rocprofv3 --kernel-trace CSV of scripts/train_bench.py (all launches).  Usage:
    python scripts/summarize_train_trace.py trace.csv > profiles/rXX_train_kernels.md"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
agg = defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "")
    n = n.replace("(anonymous namespace)::", "")
    n = n[: n.index("(")] if "(" in n else n[:70]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"# all launches: {len(rows)}, kernel time {tot/1e6:.1f} ms")
print("| kernel | launches | total ms | avg us | % |")
print("|---|---|---|---|---|")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"| `{n}` | {c} | {t/1e6:.2f} | {t/c/1e3:.1f} | {100*t/tot:.1f} |")
