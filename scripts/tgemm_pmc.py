"""Diagnostic: a few launches of the tiled GEMM at a prefill shape, to be run under rocprofv3 --pmc (one counter group per pass)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops
dev = torch.device("cuda:0")
M, N, K = int(os.environ.get("TG_M", "4800")), 2560, 2048
x = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
for _ in range(10):
    ops.tlinear(x, w)
torch.cuda.synchronize()
