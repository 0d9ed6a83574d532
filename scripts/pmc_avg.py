"""Average every counter of rocprofv3 --pmc CSVs per kernel (substring filter).  Usage: pmc_avg.py DIR [substring]"""
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:40], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
for (kn, cn), (n, t) in sorted(acc.items()):
    print(f"{kn:40s} {cn:32s} n={n:4d} avg={t/n:.4g}")
