"""Per-kernel totals of a rocprofv3 --kernel-trace CSV of scripts/train_bench.py.  Usage:
    python scripts/summarize_train_trace.py trace.csv [--one-step N] > profiles/rXX_train_kernels.md
--one-step N: only the launches of ONE steady-state step: between the cross-entropy launches (`ce_row_kernel`) of the timed steps N - 1
and N of train_bench.py (N = TB_STEPS: the last step that reuses the transposed weight copies), i.e. one backward + one forward."""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
head = f"# all launches: {len(rows)}"
if "--one-step" in sys.argv:
    ce = [i for i, r in enumerate(rows) if "ce_row_kernel" in r["Kernel_Name"]]
    N = int(sys.argv[sys.argv.index("--one-step") + 1])
    a, b = ce[N - 1], ce[N]
    rows = rows[a:b]
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    head = f"# ONE steady-state training step (between two cross-entropy launches of scripts/train_bench.py under rocprofv3 --kernel-trace): {len(rows)} launches, span {span/1e6:.1f} ms"
agg = defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "")
    n = n.replace("(anonymous namespace)::", "")
    n = n[: n.index("(")] if "(" in n else n[:70]
    agg[n][0] += 1
    agg[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"{head}, kernel time {tot/1e6:.1f} ms")
print("| kernel | launches | total ms | avg us | % of kernel time |")
print("|---|---|---|---|---|")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"| `{n[:150]}` | {c} | {t/1e6:.2f} | {t/c/1e3:.1f} | {100*t/tot:.1f} |")
