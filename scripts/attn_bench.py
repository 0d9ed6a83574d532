"""Micro-benchmark (not a test): decode attention of one layer at the headline shape (16 rows, 16 / 2 heads, hd 128) over cached lengths L,
the 8-key-split launch + merge launch against the wide form (umoe_attn_args.wide).  The K / V caches rotate over 40 sets (0.5-2 GB)
so that every call reads HBM-cold rows, as a decode step does layer after layer.  Prints one line per (form, L): us per call.

  python scripts/attn_bench.py [L ...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unimoe_audio_amd import ops

dev = torch.device("cuda:0")
rows, H, KVH, hd, NSET = 16, 16, 2, 128, 40
Ls = [int(v) for v in sys.argv[1:]] or [305, 460, 610, 814, 1600]
SPLITS = [int(v) for v in os.environ.get("AB_SPLITS", "8").split(",")]
for L in Ls:
    Lmax = ((L + 80) // 64) * 64
    g = torch.Generator().manual_seed(L)
    kcs = [(torch.randn(rows, KVH, Lmax, hd, generator=g) * 0.5).to(torch.bfloat16).to(dev) for _ in range(NSET)]
    vcs = [(torch.randn(rows, KVH, Lmax, hd, generator=g) * 0.5).to(torch.bfloat16).to(dev) for _ in range(NSET)]
    qkv = (torch.randn(rows, (H + 2 * KVH) * hd, generator=g) * 0.7).to(torch.bfloat16).to(dev)
    kv_start = torch.zeros(rows, dtype=torch.int32, device=dev)
    kv_start[0::2] = 17
    q0 = torch.full((rows,), L - 1, dtype=torch.int32, device=dev)
    cos_tab, sin_tab = ops.rope_tables(Lmax + 8, hd, 1e6, dev)
    p3 = torch.stack([q0, q0, q0]).to(torch.int32).contiguous()
    kw = dict(qkv_raw=qkv, cos_tab=cos_tab, sin_tab=sin_tab, pos3=p3, sections=(16, 24, 24))
    forms = [(f"split{n}+merge", dict(splits=n)) for n in SPLITS] + [("wide", dict(splits=1, wide=1))]
    for name, akw in forms:
        def run(n):
            for i in range(n):
                ops.attention(None, kcs[i % NSET], vcs[i % NSET], kv_start, q0, 1, H, **akw, **kw)
        run(NSET)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        # capture the rotation in a graph: launch gaps as in the decode step graph, no host time in the figure
        gph = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            run(NSET)
            torch.cuda.synchronize()
            with torch.cuda.graph(gph, stream=s):
                run(NSET)
        torch.cuda.synchronize()
        gph.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            gph.replay()
        e1.record()
        torch.cuda.synchronize()
        print(f"L {L:5d}  {name:13s} {e0.elapsed_time(e1) * 1e3 / (5 * NSET):7.2f} us per layer-call")
    del kcs, vcs
