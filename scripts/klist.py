"""Diagnostic: durations of every launch of the kernels whose name contains a substring, in launch order.  Usage: klist.py trace.csv substring"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join(f'{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:.0f}' for r in rows))
