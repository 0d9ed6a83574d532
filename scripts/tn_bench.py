"""Weight-gradient products of one training layer (BASELINE configs[2]: 6240 tokens, ~3.45 routed experts per token): umoe_tiled_gemm_tn on
the row-major activations against the round-2 path (umoe_transpose_slots copies + umoe_tiled_gemm), one process, interleaved rounds.
Prints one line per product: microseconds and TFLOP/s of both, and whether the outputs are bit-identical."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)


def rnd(*shape):
    return (torch.randn(*shape, generator=g) * 0.5).to(torch.bfloat16).to(dev)


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def report(name, flop, t_tn, t_nt, t_tr, same):
    print(f"{name:34s} tn {t_tn:7.1f} us {flop / t_tn * 1e-6:6.0f} TF/s | nt {t_nt:7.1f} us {flop / t_nt * 1e-6:6.0f} TF/s + transposes {t_tr:6.1f} us | "
          f"tn vs nt+tr {(t_nt + t_tr) / t_tn:5.2f}x  bit-identical {same}", flush=True)


def static_case(name, T, M, N, G=1, ksplits=(1,)):
    P, Q = rnd(G * T, M), rnd(G * T, N)
    out = torch.empty(G * M, N, dtype=torch.bfloat16, device=dev)
    nt = torch.empty(G * M, N, dtype=torch.bfloat16, device=dev)
    flop = 2.0 * G * T * M * N
    Tp = (T + 7) & ~7
    PT = torch.empty(M, G * Tp, dtype=torch.bfloat16, device=dev)
    QT = torch.empty(N, G * Tp, dtype=torch.bfloat16, device=dev)

    def tr():
        for i in range(G):
            ops.transpose_slots(P[i * T:(i + 1) * T], PT[:, i * Tp:(i + 1) * Tp])
            ops.transpose_slots(Q[i * T:(i + 1) * T], QT[:, i * Tp:(i + 1) * Tp])

    def run_nt():
        ops.tiled_gemm([dict(w=QT[:, i * Tp:(i + 1) * Tp], static_count=M, out_row_base=i * M, a_col_off=i * Tp, k=Tp) for i in range(G)], PT, nt, max_rows=M)

    tr()
    run_nt()
    t_tr, t_nt = timeit(tr), timeit(run_nt)
    for ks in ksplits:
        groups = [dict(m=M, n=N, k_off=i * T, k=T, out_row_base=i * M) for i in range(G)]
        f = lambda: ops.tiled_gemm_tn(groups, P, Q, out, k_split=ks)
        f()
        same = bool(torch.equal(out, nt))
        report(f"{name} T={T} {G}x{M}x{N} ks={ks}", flop, timeit(f), t_nt, t_tr, same)


def routed_case(name, counts, M, N, PC, p_col_off=0):
    E = len(counts)
    offs, tot = [], 0
    for c in counts:
        offs.append(tot)
        tot += (c + 7) & ~7
    P, Q = rnd(tot, PC), rnd(tot, N)
    cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
    off = torch.tensor(offs, dtype=torch.int32, device=dev)
    out = torch.empty(E * M, N, dtype=torch.bfloat16, device=dev)
    groups = [dict(m=M, n=N, p_col_off=p_col_off, k_off_dev=off[e:e + 1], k_count_dev=cnt[e:e + 1], out_row_base=e * M) for e in range(E)]
    f = lambda: ops.tiled_gemm_tn(groups, P, Q, out)
    flop = 2.0 * sum(counts) * M * N
    t = timeit(f)
    print(f"{name:34s} tn {t:7.1f} us {flop / t * 1e-6:6.0f} TF/s   ({E} experts, {sum(counts)} slots, {M}x{N})", flush=True)


if __name__ == "__main__":
    static_case("dense QKV dW", 6240, 2560, 2048, ksplits=(1, 2, 3, 4))
    static_case("dense o_proj dW", 6240, 2048, 2048, ksplits=(1, 2, 3, 4))
    static_case("shared gate|up dW", 6240, 2752, 2048, G=2, ksplits=(1, 2, 3))
    static_case("shared down dW", 6240, 2048, 1376, G=2, ksplits=(1, 2, 3, 4))
    counts = [2700, 2300, 3111, 2508, 2901, 2999, 2600, 2427]     # 21 546 slots: 6240 tokens x 3.45
    routed_case("routed gate dW (half of dG|dU)", counts, 2752, 2048, 5504)
    routed_case("routed down dW", counts, 2048, 2752, 2048)
