"""Race screen of the three tile GEMM kernels after a change of their barrier schedule (cdna_hip_programming.md: a sync-structure edit makes a
new template -- screen it over many runs at several sizes): every shape runs N times, alone and beside a second stream that keeps the chip
busy with other GEMMs; every output must equal the first run's bit for bit (the kernels are deterministic), and the first run must match an
fp32 reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(7)
rnd = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(torch.bfloat16).to(dev)
side = torch.cuda.Stream()
noise_x, noise_w = rnd(4096, 1024), rnd(2048, 1024)
bad = 0
shapes = [(6240, 2560, 2048), (1100, 520, 96), (2304, 3584, 704), (300, 2048, 12324), (4099, 264, 40), (2600, 4000, 520)]
for (S, Nn, K) in shapes:
    K8 = (K + 7) & ~7
    x = torch.zeros(S, K8, dtype=torch.bfloat16, device=dev)
    x[:, :K] = rnd(S, K)
    w_nt, w_kmaj = rnd(Nn, K8), rnd(K, Nn)
    w_nt[:, K:] = 0
    P, Q = rnd(S, (Nn + 7) & ~7), rnd(S, K8)
    outs = {}
    for name in ("nt", "kmajor", "tn"):
        ref = None
        for it in range(N):
            busy = it % 2 == 1
            if busy:
                with torch.cuda.stream(side):
                    for _ in range(3):
                        ops.tlinear(noise_x, noise_w)
            if name == "nt":
                y = ops.tlinear(x, w_nt)
            elif name == "kmajor":
                y = torch.empty(S, Nn, dtype=torch.bfloat16, device=dev)
                ops.tiled_gemm([dict(w=w_kmaj, w_kmajor=1, static_count=S)], x, y, max_rows=S)
            else:
                y = torch.empty(Nn, K8, dtype=torch.bfloat16, device=dev)
                ops.tiled_gemm_tn([dict(m=Nn, n=K8, k=S)], P, Q, y)
            if ref is None:
                ref = y.clone()
                if name == "nt":
                    want = x.float() @ w_nt.float().t()
                elif name == "kmajor":
                    want = x[:, :K].float() @ w_kmaj.float()
                else:
                    want = P[:, :Nn].float().t() @ Q.float()
                err = float((ref.float() - want).abs().max() / (want.abs().max() + 1e-9))
                assert err < 2 ** -6, (name, S, Nn, K, err)
            elif not torch.equal(y, ref):
                bad += 1
                print(f"MISMATCH {name} {S}x{Nn}x{K} iteration {it} (busy={busy}): {int((y != ref).sum())} elements differ", flush=True)
        torch.cuda.synchronize()
    print(f"{S}x{Nn}x{K}: {N} runs of each kernel", flush=True)
print("race screen:", "CLEAN" if bad == 0 else f"{bad} mismatching runs")
sys.exit(1 if bad else 0)
