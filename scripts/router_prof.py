import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
S, D = 16, 2048
x = torch.randn(S, D, device=dev).to(torch.bfloat16)
gw = (torch.randn(11, D, device=dev) * 0.02).to(torch.bfloat16)
nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
lg = (torch.randn(S, 11, device=dev) * 0.9).to(torch.bfloat16)
N = 20
for _ in range(N): ops.router_dispatch_fwd(x, gw, n_dyn=9, n_real=8, n_fix=2, top_p=0.7, norm_w=nw)          # A fused full
torch.cuda.synchronize()
for _ in range(N): ops.router_fwd(None, None, n_dyn=9, n_real=8, n_fix=2, top_p=0.7, logits_in=lg)             # B routing only
torch.cuda.synchronize()
for _ in range(N): ops.router_fwd(None, None, n_dyn=9, n_real=8, n_fix=2, top_p=0.0, fixed_top_k=1, logits_in=lg)  # C 1 round, no top-p
torch.cuda.synchronize()
for _ in range(N): ops.router_fwd(None, None, n_dyn=9, n_real=8, n_fix=2, top_p=0.0, fixed_top_k=9, logits_in=lg)  # D 9 rounds
torch.cuda.synchronize()
for _ in range(N): ops.router_fwd(x, gw, n_dyn=9, n_real=8, n_fix=2, top_p=0.0, fixed_top_k=1, norm_w=nw, want_h=True)  # E gemv + 1 round (global path)
torch.cuda.synchronize()
