"""Print the rows of a rocprofv3 kernel_stats CSV whose kernel name contains a substring.  Usage: kstat.py DIR substring"""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"]:
            n = r["Name"].replace("void ", "")
            print(f'{n[:n.index("(")] if "(" in n else n[:40]:34s} calls {r["Calls"]:>4s}  avg {float(r["AverageNs"])/1e3:8.1f} us')
