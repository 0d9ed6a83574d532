"""Diagnostic: a few launches of the weight-gradient GEMM (umoe_tiled_gemm_tn, routed-expert shape with device windows) and of the forward
GEMM beside it, to be run under rocprofv3 --pmc (one counter group per pass): LDS bank conflicts of the transposing reads, MFMA busy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops
dev = torch.device("cuda:0")
counts = [2700, 2300, 3111, 2508, 2901, 2999, 2600, 2427]
offs, tot = [], 0
for c in counts:
    offs.append(tot)
    tot += (c + 7) & ~7
I, D = 2752, 2048
P = (torch.randn(tot, 2 * I, device=dev) * 0.5).to(torch.bfloat16)
Q = (torch.randn(tot, D, device=dev) * 0.5).to(torch.bfloat16)
cnt = torch.tensor(counts, dtype=torch.int32, device=dev)
off = torch.tensor(offs, dtype=torch.int32, device=dev)
out = torch.empty(8 * I, D, dtype=torch.bfloat16, device=dev)
groups = [dict(m=I, n=D, k_off_dev=off[e:e + 1], k_count_dev=cnt[e:e + 1], out_row_base=e * I) for e in range(8)]
x = (torch.randn(6240, 2048, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(2560, 2048, device=dev) * 0.02).to(torch.bfloat16)
dy = (torch.randn(6240, 2560, device=dev) * 0.5).to(torch.bfloat16)
dx = torch.empty(6240, 2048, dtype=torch.bfloat16, device=dev)
for _ in range(5):
    ops.tiled_gemm_tn(groups, P, Q, out)
    ops.tlinear(x, w)
    ops.tiled_gemm([dict(w=w, w_kmajor=1, static_count=6240)], dy, dx, max_rows=6240)      # dX = dY W on the weight as stored
torch.cuda.synchronize()
