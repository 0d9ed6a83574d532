"""Diagnostic: the grouped weight-gradient product of the training step (8 experts, dW_e [2048][2752] = dyT[:, win_e] . hT[:, win_e]^T,
contraction over the expert's ~2810 token slots) in three operand layouts.
  wide:    one [rows][25k] buffer per operand, per-expert column windows (what umoe_bwd.hip builds today)
  compact: per-expert [rows][2816] buffers (row stride = the window)
  dense:   one [2048][2816] x [2752][2816] product x 8 launches (reference point)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops
dev = torch.device("cuda:0")
G, D, I, Kw = 8, 2048, 2752, 2816
tot = G * Kw
bf = torch.bfloat16
dyT = torch.randn(D, tot, device=dev).to(bf)
hT = torch.randn(I, tot, device=dev).to(bf)
out = torch.empty(G * D, I, device=dev, dtype=bf)
offs = torch.arange(G, device=dev, dtype=torch.int32) * Kw
cnts = torch.full((G,), 2810, device=dev, dtype=torch.int32)


def timeit(fn, it=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / it * 1e6


fl = 2.0 * G * D * I * 2810


def wide():
    groups = [dict(w=hT, k=8, k_off=offs[g:g + 1], k_count=cnts[g:g + 1], static_count=D, out_row_base=g * D) for g in range(G)]
    ops.tiled_gemm(groups, dyT, out, max_rows=D)


def wide_static():
    groups = [dict(w=hT[:, g * Kw:(g + 1) * Kw], k=Kw, a_col_off=g * Kw, static_count=D, out_row_base=g * D) for g in range(G)]
    ops.tiled_gemm(groups, dyT, out, max_rows=D)


dyC = [dyT[:, g * Kw:(g + 1) * Kw].contiguous() for g in range(G)]
hC = [hT[:, g * Kw:(g + 1) * Kw].contiguous() for g in range(G)]
dyCs = torch.stack(dyC, 0).reshape(G * D, Kw).contiguous()      # [G*D][Kw]: group g's rows at a_row_base g*D


def compact():
    groups = [dict(w=hC[g], k=Kw, static_count=D, a_row_base=g * D, out_row_base=g * D) for g in range(G)]
    ops.tiled_gemm(groups, dyCs, out, max_rows=D)


def dense8():
    for g in range(G):
        ops.tiled_gemm([dict(w=hC[g], static_count=D)], dyC[g], out[g * D:(g + 1) * D], max_rows=D)


for name, fn in (("wide window (k_off / k_count on device)", wide), ("wide window (static a_col_off)", wide_static), ("compact per-expert buffers, one launch", compact),
                 ("compact, 8 launches", dense8)):
    t = timeit(fn)
    print(f"{name}: {t:.1f} us  {fl / t / 1e6:.0f} TFLOP/s", flush=True)

# ---- the ragged-ROW grouped products of the same layer (forward down projection: rows of every expert, N = 2048, K = 2752) ----
S = 6240
cnt_list = [2810] * G
tot_rows = sum((c + 7) & ~7 for c in cnt_list)
hbuf = torch.randn(tot_rows, I, device=dev).to(bf)
wd = [torch.randn(D, I, device=dev).to(bf) for _ in range(G)]
yb = torch.empty(tot_rows, D, device=dev, dtype=bf)
counts = torch.tensor(cnt_list, device=dev, dtype=torch.int32)
offsets = torch.tensor([sum((c + 7) & ~7 for c in cnt_list[:g]) for g in range(G)], device=dev, dtype=torch.int32)
fl2 = 2.0 * sum(cnt_list) * D * I


def ragged_rows(max_rows):
    groups = [dict(w=wd[g], count=counts[g:g + 1], row_off=offsets[g:g + 1]) for g in range(G)]
    ops.tiled_gemm(groups, hbuf, yb, max_rows=max_rows)


def static_rows():
    groups = [dict(w=wd[g], static_count=cnt_list[g], a_row_base=int(offsets[g]), out_row_base=int(offsets[g])) for g in range(G)]
    ops.tiled_gemm(groups, hbuf, yb, max_rows=max(cnt_list))


for name, fn in (("down fwd, ragged rows, max_rows 6240", lambda: ragged_rows(S)), ("down fwd, ragged rows, max_rows 2816", lambda: ragged_rows(2816)),
                 ("down fwd, static rows", static_rows)):
    t = timeit(fn)
    print(f"{name}: {t:.1f} us  {fl2 / t / 1e6:.0f} TFLOP/s", flush=True)
