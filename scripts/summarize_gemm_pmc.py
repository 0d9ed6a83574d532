"""Per-launch averages of rocprofv3 --pmc passes (one counter per pass, directories /tmp/pmc_*) of scripts/tn_pmc.py for the tiled GEMM kernels."""
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "tgemm" not in n:
            continue
        n = n.replace("void ", "")
        n = n[: n.index("(")]
        k = (n, r["Counter_Name"])
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
print("# rocprofv3 --pmc (one counter per pass) of scripts/tn_pmc.py: per-launch averages; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)")
for n in sorted({k[0] for k in acc}):
    g = lambda c: acc[(n, c)][1] / max(1, acc[(n, c)][0])
    busy = g("SQ_VALU_MFMA_BUSY_CYCLES") / max(1.0, 4 * g("SQ_BUSY_CU_CYCLES"))
    print(f"{n}: SQ_LDS_BANK_CONFLICT {g('SQ_LDS_BANK_CONFLICT'):.0f}  SQ_VALU_MFMA_BUSY_CYCLES {g('SQ_VALU_MFMA_BUSY_CYCLES'):.0f}  "
          f"SQ_BUSY_CU_CYCLES {g('SQ_BUSY_CU_CYCLES'):.0f}  MFMA busy {busy:.3f}")
