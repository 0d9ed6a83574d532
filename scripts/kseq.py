"""Diagnostic: launch sequence (short names, duration us, gap before us) of the last `n` launches of a kernel trace.  Usage: kseq.py trace.csv n"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[2]):]
prev = None
for r in rows:
    n = r["Kernel_Name"].replace("void ", "")
    m = re.findall(r"at::native::(?:\(anonymous namespace\)::)?([A-Za-z_0-9]+)", n)
    short = ("torch:" + "/".join(dict.fromkeys(m[:3]))) if m else (n[: n.index("(")] if "(" in n else n[:50])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    gap = (int(r["Start_Timestamp"]) - prev) / 1e3 if prev else 0
    prev = int(r["End_Timestamp"])
    print(f"{d:8.1f} {gap:7.1f}  {short[:90]}")
