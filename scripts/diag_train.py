"""Diagnostic (not a test): per-parameter gradient errors of the HIP training step against the CPU autograd oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_engine as TE
from unimoe_audio_amd import train as TR
from oracle import train_autograd as OT
dev = torch.device("cuda:0")
cfg = TE.small_cfg()
m, w = TE.build(cfg, 3, 0.05)
B, T = 2, 40
ids, am, codec = TE.prompt(cfg, 1, T, 5, [6, 0])
torch.manual_seed(9)
labels = torch.randint(0, 1024, (B, T, cfg.codec_channels))
labels[:, :8] = -100
labels[:, :, 11] = -100
gm = m.to(dev).train()
for p_ in gm.parameters():
    p_.requires_grad_(True)
auxw = float(gm.cur_aux_weight)
xnoise = TE.inject_input_jitter(cfg, gm, B, T, 77)
loss, closs, auxm, routing = TR.forward_train(gm, ids, codec, am, labels, return_routing=True)
loss.backward()
wo = {k: v.clone().requires_grad_(True) for k, v in w.items()}
forced = [(k_.cpu(), m_.cpu()) for k_, m_ in routing]
lo, clo, auxo, hs = OT.forward_loss(cfg, wo, ids, codec, am, labels, auxw, training=True, forced=forced, input_noise=xnoise)
lo.backward()
print("loss", float(loss), float(lo), "aux", float(auxm), float(auxo))
rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
rows = []
for n, p_ in gm.named_parameters():
    ref = wo[n].grad
    if ref is None or float(ref.float().norm()) == 0:
        continue
    rows.append((rel(p_.grad.cpu(), ref), float(ref.float().norm()), n))
for r in sorted(rows, reverse=True):
    print(f"{r[0]:.4f}  |ref|={r[1]:.3e}  {r[2]}")
print("median", sorted(r[0] for r in rows)[len(rows) // 2])
