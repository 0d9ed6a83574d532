"""Diagnostic: GPU busy fraction of the steady part of a kernel trace (from the first router launch on).
Usage: busy.py trace.csv [last_ms]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = next(i for i, r in enumerate(rows) if "router_kernel" in r["Kernel_Name"])      # first forward pass
rows = rows[first:]
if len(sys.argv) > 2:                                    # only the last N ms of the trace (steady state)
    t_end = int(rows[-1]["End_Timestamp"])
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t_end - int(float(sys.argv[2]) * 1e6)]
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
gaps = sorted((int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])), reverse=True)
print(f"{len(rows)} launches, span {span/1e6:.1f} ms, busy {busy/1e6:.1f} ms = {100*busy/span:.1f} %; largest gaps (us): {[round(g/1e3) for g in gaps[:8]]}; "
      f"gaps > 20 us: {sum(1 for g in gaps if g > 20000)} totalling {sum(g for g in gaps if g > 20000)/1e6:.1f} ms")
