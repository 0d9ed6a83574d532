"""Diagnostic (not the product path): cycle stamps of one K tile (one load + one MFMA segment) of the ping-pong tiled GEMM, workgroup 0,
all 8 waves.  Needs the instrumented build (make -C unimoe_audio_amd/csrc tl: -DUMOE_PP_STAMPS)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unimoe_audio_amd", "csrc")
subprocess.check_call(["make", "-C", CSRC, "tl", "-j4", "-s"])
os.environ["UMOE_HIP_LIB"] = os.path.join(CSRC, "libumoe_hip_tl.so")
os.environ["UMOE_TGEMM_PP"] = "1"
sys.path.insert(0, ROOT)
import ctypes as C
import torch
from unimoe_audio_amd import _lib, ops
dev = torch.device("cuda:0")
L = _lib.lib()
tl = torch.zeros(64 * 16 * 16 + 8, dtype=torch.int64, device=dev)      # the timeline build needs its buffer installed
for name in ("gemm", "router", "attn", "misc"):
    fn = getattr(L, "umoe_tl_set_" + name)
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    fn(tl.data_ptr())
M, N, K = 6240, 2560, 2048
x = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
aux = torch.zeros(8 * 16 * 4, dtype=torch.bfloat16, device=dev)           # 8 waves x 10 stamps x 8 bytes
for _ in range(3):
    ops.tiled_gemm([dict(w=w, static_count=M)], x, out, max_rows=M, epilogue=ops.EPI_BF16, aux_out=aux)
torch.cuda.synchronize()
st = aux.view(torch.int64).cpu().reshape(8, 16)
t0 = int(st[:, 0].min())
# (one load segment + one MFMA segment of 32 per K tile since the end of round 3: stamps 4-7 of the two-phase form are unused and read 0)
names = ["L start", "L issued", "waits done", "after bar1", "-", "-", "-", "-", "M issued", "after bar2"]
print("cycles relative to the earliest wave's phase start (shader clock); one row per wave (0-3 group 0, 4-7 group 1)")
print(" " * 8 + " ".join(f"{n[:12]:>13s}" for n in names))
for wv in range(8):
    print(f"wave {wv}: " + " ".join(f"{int(st[wv, k]) - t0:13d}" for k in range(10)) + f"   entry->loop {int(st[wv,10]-st[wv,12])} loop {int(st[wv,11]-st[wv,10])} loop->exit {int(st[wv,13]-st[wv,11])} | ticks/us {int(st[wv,13]-st[wv,12]) / max((int(st[wv,15]-st[wv,14])) / 100.0, 1e-9):.0f}")
