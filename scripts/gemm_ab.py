"""A/B of two builds of the library on ONE tiled-GEMM shape (run once per build, alternating, on one box):
   UMOE_HIP_LIB=... python scripts/gemm_ab.py [rows=6240] [n=2560] [k=2048] [iters=300]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops
KMAJ = "--kmajor" in sys.argv          # Y = A W with the weight's row as the contraction index (the input-gradient form)
argv = [a for a in sys.argv[1:] if a != "--kmajor"]
S, N, K, it = [int(v) for v in (argv + ["6240", "2560", "2048", "300"][len(argv):])]
dev = torch.device("cuda:0")
x = (torch.randn(S, K, device=dev) * 0.5).to(torch.bfloat16)
ws = [(torch.randn(*((K, N) if KMAJ else (N, K)), device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
b = torch.zeros(N, device=dev)
out = torch.empty(S, N, dtype=torch.bfloat16, device=dev)
if KMAJ:
    run = lambda i: ops.tiled_gemm([dict(w=ws[i % 4], w_kmajor=1, static_count=S)], x, out, max_rows=S)
else:
    run = lambda i: ops.tlinear(x, ws[i % 4], bias=b)
for i in range(10):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for r in range(5):
    e0.record()
    for i in range(it):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / it * 1e3)
print(os.environ.get("UMOE_HIP_LIB", "default"), ("kmajor " if KMAJ else "") + f"{S}x{N}x{K}", [round(t, 2) for t in ts], "us; best", round(2.0 * S * N * K / min(ts) * 1e-6), "TFLOP/s")
