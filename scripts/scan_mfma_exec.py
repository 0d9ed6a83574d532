"""Guard against a code-generation hazard: MFMA ignores EXEC.  When a guard around an MFMA is a per-lane condition, hipcc may
predicate the short block through EXEC (s_and_saveexec) WITHOUT a skip branch; the MFMA then runs anyway (seen in umoe_gemm.hip:
the guarded last k-step of a partial chunk was counted twice).  Every such guard must be wave-uniform in SGPRs (readfirstlane).
Usage: scan_mfma_exec.py file.s [...] -> exit 1 when an MFMA directly follows s_and_saveexec."""
import sys


def scan(path):
    lines = open(path).read().split("\n")
    bad, cur = [], None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.endswith(":"):
            cur = l[:-1]
        if "v_mfma" in l:
            k = i - 1
            while k >= 0 and (not lines[k].strip() or lines[k].strip().startswith(";") or lines[k].strip().endswith(":")):
                k -= 1
            if "saveexec" in lines[k]:
                bad.append((cur, i + 1))
    return sum("v_mfma" in l for l in lines), bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        n, bad = scan(p)
        print(f"{p}: {n} MFMA instructions, {len(bad)} under s_and_saveexec without a skip branch {bad[:3]}")
        rc |= bool(bad)
    sys.exit(rc)
