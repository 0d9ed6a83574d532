"""Summarise rocprofv3 --pmc passes (one counter per pass) into per-kernel averages for the DECODE launches.
Usage: python scripts/summarize_pmc.py FETCH_DIR WRITE_DIR > profiles/rXX_pmc_traffic.md   (also writes the .json beside it
when a third argument names it).  Counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced
streaming reads (MI355X_MICROARCH.md, HBM section): fetch bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact."""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            n = r["Kernel_Name"].replace("void ", "")
            n = n[: n.index("(")] if "(" in n else n[:70]
            key = (n, int(r["Grid_Size"]))
            acc[key][0] += 1
            acc[key][1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
alg = {"moe_fused_kernel": 304.3, "wstream_gemm_rk<0, 2>": 10.5, "wstream_gemm_rk<1, 3>": 8.4, "wstream_gemm<1, 16, 0, 0, 4, false, false, true>": 10.5,
       "wstream_gemm<1, 16, 0, 1, 4, false, false, true>": 8.4, "wstream_gemm<2, 8, 0, 3, 4, false, false, true>": 50.5,
       "wstream_gemm<14, 1, 0, 2, 8, true, true>": 202.9, "wstream_gemm<14, 1, 0, 2, 8, true, false>": 202.9, "wstream_gemm<6, 2, 0, 0, 8, false, false>": 101.4,
       "wstream_gemm<1, 16, 0, 0, 4, false, false>": 10.5, "wstream_gemm<1, 16, 0, 1, 4, false, false>": 8.4, "wstream_gemm<2, 8, 0, 3, 4, false, false>": 50.5,
       "wstream_gemm<14, 1, 0, 2, 8, true>": 202.9, "wstream_gemm<14, 1, 0, 2, 8, false>": 202.9, "wstream_gemm<6, 2, 0, 0, 8, false>": 101.4,
       "wstream_gemm<1, 16, 0, 0, 4, false>": 10.5, "wstream_gemm<1, 16, 0, 1, 4, false>": 8.4, "wstream_gemm<2, 8, 0, 3, 4, false>": 50.5,
       "wstream_gemm<14, 1, 0, 2, 8>": 202.9, "wstream_gemm<8, 2, 0, 2, 4>": 202.9, "wstream_gemm<6, 2, 0, 0, 8>": 101.4, "wstream_gemm<8, 2, 0, 0, 8>": 101.4, "wstream_gemm<1, 16, 0, 0, 4>": 10.5,
       "wstream_gemm<1, 16, 0, 1, 4>": 8.4, "wstream_gemm<2, 8, 0, 3, 4>": 50.5}
rows = []
for key, (cnt, tot) in fetch.items():
    if cnt < 100:          # decode launches repeat every layer of every step
        continue
    w = write.get(key, [1, 0.0])
    f_kib, w_kib = tot / cnt, w[1] / max(1, w[0])
    rows.append(dict(kernel=key[0], grid_threads=key[1], launches=cnt, fetch_kib_raw=round(f_kib, 1), fetch_mb=round(f_kib * 1024 * 2 / 1e6, 1),
                     write_kib=round(w_kib, 1), traffic_mb=round((f_kib * 2048 + w_kib * 1024) / 1e6, 1), algorithmic_mb=alg.get(key[0])))
rows.sort(key=lambda r: -r["traffic_mb"])
print("| kernel (decode grid) | launches | FETCH_SIZE KiB (raw) | fetch MB (x2) | WRITE_SIZE KiB | traffic MB | algorithmic MB |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| `{r['kernel']}` ({r['grid_threads']} threads) | {r['launches']} | {r['fetch_kib_raw']} | {r['fetch_mb']} | {r['write_kib']} | "
          f"{r['traffic_mb']} | {r['algorithmic_mb'] or ''} |")
if len(sys.argv) > 3:     # schema read by bench.py: {"kernels": {name: {"traffic_bytes": ...}}}
    out = {"method": __doc__, "kernels": {}}
    for r in rows:
        out["kernels"].setdefault(r["kernel"], {"grid_threads": r["grid_threads"], "launches": r["launches"],
                                                "fetch_size_kib_raw": r["fetch_kib_raw"], "fetch_bytes_x2": int(r["fetch_kib_raw"] * 2048),
                                                "write_bytes": int(r["write_kib"] * 1024),
                                                "traffic_bytes": int(r["fetch_kib_raw"] * 2048 + r["write_kib"] * 1024),
                                                "algorithmic_mb": r["algorithmic_mb"]})
    json.dump(out, open(sys.argv[3], "w"), indent=1)
