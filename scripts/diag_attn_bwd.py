"""Diagnostic: fused attention backward vs CPU autograd, error per part (dq / dk / dv) and per token block."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from unimoe_audio_amd import ops, train as TR
from oracle import decode as OD
dev = torch.device("cuda:0")
rel = lambda a, b: float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))
for (B, T, H, KVH) in [(1, 64, 2, 2), (1, 65, 2, 2), (1, 80, 2, 2), (1, 130, 2, 2)]:
    g = torch.Generator().manual_seed(6 + T)
    hd = 128
    D = H * hd
    sections = [16, 24, 24]
    qkv = (torch.randn(B * T, (H + 2 * KVH) * hd, generator=g) * 0.7).to(torch.bfloat16)
    G_ = torch.randn(B * T, D, generator=g).to(torch.bfloat16)
    am = torch.ones(B, T, dtype=torch.long)
    am[0, :7] = 0
    pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
    x = qkv.clone().requires_grad_(True)
    q = x[:, :D].view(B, T, H, hd).transpose(1, 2)
    k = x[:, D:D + KVH * hd].view(B, T, KVH, hd).transpose(1, 2)
    v = x[:, D + KVH * hd:].view(B, T, KVH, hd).transpose(1, 2)
    cos3, sin3 = OD.rope_cos_sin(pos[None].expand(3, -1, -1), hd, 1000000.0, torch.bfloat16)
    cos, sin = OD.mrope_select(cos3, sections), OD.mrope_select(sin3, sections)
    qr, kr = OD.apply_rope(q, k, cos, sin)
    gq = H // KVH
    sc = torch.matmul(qr.float(), kr.float().repeat_interleave(gq, 1).transpose(2, 3)) * hd ** -0.5
    allowed = (torch.arange(T).view(1, 1, 1, T) <= torch.arange(T).view(1, 1, T, 1)) & am.bool().view(B, 1, 1, T)
    p = torch.nan_to_num(torch.softmax(sc.masked_fill(~allowed, float("-inf")), -1), nan=0.0)
    o = torch.matmul(p, v.float().repeat_interleave(gq, 1)).to(torch.bfloat16).transpose(1, 2).reshape(B * T, D)
    valid = am.bool().reshape(-1)
    (o.float() * G_.float() * valid[:, None]).sum().backward()
    xg = qkv.to(dev).requires_grad_(True)
    cosd, sind = ops.rope_tables(int(pos.max()) + 2, hd, 1000000.0, dev)
    pos3 = pos[None].expand(3, -1, -1).reshape(3, B * T).to(torch.int32).contiguous().to(dev)
    kv_pos = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
    fv = (am != 0).float().argmax(-1).to(torch.int32)
    ao = TR.RopeAttentionFn.apply(xg, cosd, sind, pos3, kv_pos, fv.to(dev), fv.tolist(), B, T, H, KVH, hd, tuple(sections))
    (ao.float() * (G_.to(dev).float() * valid.to(dev)[:, None])).sum().backward()
    gg, gr = xg.grad.cpu(), x.grad
    print(f"B{B} T{T} H{H} KVH{KVH}: fwd {rel(ao.detach().cpu()[valid], o.detach()[valid]):.4f}  dq {rel(gg[valid][:, :D], gr[valid][:, :D]):.4f}"
          f"  dk {rel(gg[valid][:, D:D + KVH * hd], gr[valid][:, D:D + KVH * hd]):.4f}  dv {rel(gg[valid][:, D + KVH * hd:], gr[valid][:, D + KVH * hd:]):.4f}")
    for blk in range(0, T, 64):
        sl = slice(blk, min(blk + 64, T))
        vv = valid[sl]
        print(f"   tokens {blk:4d}..: dq {rel(gg[sl][vv][:, :D], gr[sl][vv][:, :D]):.4f} dk {rel(gg[sl][vv][:, D:D + KVH * hd], gr[sl][vv][:, D:D + KVH * hd]):.4f}"
              f" dv {rel(gg[sl][vv][:, D + KVH * hd:], gr[sl][vv][:, D + KVH * hd:]):.4f}")

    dvg = gg[:, D + KVH * hd:].float().view(T, KVH, hd)
    dvr = gr[:, D + KVH * hd:].float().view(T, KVH, hd)
    print("   dv err per 16-key group (kvh 0):", [round(rel(dvg[i:i + 16, 0][valid[i:i+16]], dvr[i:i + 16, 0][valid[i:i+16]]), 3) for i in range(0, T, 16)])
    print("   dv err per d-block (keys 8..63):", [round(rel(dvg[8:64, 0, j:j + 16], dvr[8:64, 0, j:j + 16]), 3) for j in range(0, 128, 16)])
    ratio = (dvg[8:64, 0] / dvr[8:64, 0]).flatten()
    print("   median ratio gpu/ref:", float(ratio.median()))
