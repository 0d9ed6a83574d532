"""GPU parity tests of the backward entry points, one by one, against torch autograd on the CPU (fp32 reference of the same
op with the reference's rounding points) or against the differentiable oracle.  Tolerances are relative Frobenius errors
of bf16 results (stated per test)."""
import types

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from unimoe_audio_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def rel(a, b):
    return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-12))


def test_transpose_slots_plain_and_grouped(dev):
    from unimoe_audio_amd import ops
    g = torch.Generator().manual_seed(1)
    src = torch.randn(203, 130, generator=g).to(torch.bfloat16)
    t = ops.transpose(src.to(dev)).cpu()
    assert t.shape == (130, 208)
    assert torch.equal(t[:, :203], src.t()) and float(t[:, 203:].abs().sum()) == 0.0          # zero padded to 8
    # whole 16-byte chunks (C % 8 == 0, the vector path of the tile kernel): ragged row counts around its tile edges, a column view of
    # a wider source.  (A register transpose -- 8 x 8 blocks per thread, no LDS -- was measured against this kernel: 3.6-4.2 TB/s
    # against 3.6-4.9 TB/s read + write at the training shapes; not kept)
    for R, Cc in ((203, 136), (1, 8), (8, 8), (129, 128), (257, 264), (1560, 2048)):
        src = torch.randn(R, Cc, generator=g).to(torch.bfloat16)
        t = ops.transpose(src.to(dev)).cpu()
        assert t.shape == (Cc, (R + 7) // 8 * 8)
        assert torch.equal(t[:, :R], src.t()) and float(t[:, R:].abs().sum()) == 0.0, (R, Cc)
    wide = torch.randn(77, 200, generator=g).to(torch.bfloat16).to(dev)
    view = wide[:, 40:168]                                                # 128 columns starting at a multiple of 8, row stride 200
    t = torch.empty((128, 80), dtype=torch.bfloat16, device=dev)
    ops.transpose_slots(view, t)
    assert torch.equal(t.cpu()[:, :77], view.cpu().t()) and float(t.cpu()[:, 77:].abs().sum()) == 0.0
    # grouped: 3 experts, 8-aligned slot ranges, gather list
    mask = (torch.rand(90, 5, generator=g) < 0.4).to(torch.int32)
    mask[:, 1] = 0
    d = ops.dispatch_build_aligned(mask.to(dev), 3, 8)
    offs, cnts, st = d["offsets"].cpu(), d["counts"].cpu(), d["slot_token"].cpu()
    assert all(int(offs[e]) % 8 == 0 for e in range(4)) and int(cnts[1]) == 0
    x = torch.randn(90, 72, generator=g).to(torch.bfloat16)
    ld = (int(offs[3]) + 7) // 8 * 8
    dst = torch.full((72, ld), 7.0, dtype=torch.bfloat16, device=dev)
    ops.transpose_slots(x.to(dev), dst, rows=d["slot_token"], counts=d["counts"], offsets=d["offsets"], n_groups=3, max_rows=90)
    dst = dst.cpu()
    for e in range(3):
        o, c = int(offs[e]), int(cnts[e])
        rows = torch.nonzero(mask[:, e]).flatten()
        assert torch.equal(st[o:o + c].long(), rows)
        assert torch.equal(dst[:, o:o + c], x[rows].t())
        pad = (c + 7) // 8 * 8
        assert float(dst[:, o + c:o + pad].abs().sum()) == 0.0


def test_swiglu_bwd_vs_autograd(dev):
    from unimoe_audio_amd import ops
    g = torch.Generator().manual_seed(2)
    R, I = 77, 96
    gu = (torch.randn(R, 2 * I, generator=g) * 1.5).to(torch.bfloat16)
    dh = torch.randn(R, I, generator=g).to(torch.bfloat16)
    gg, uu = gu[:, :I].clone().requires_grad_(True), gu[:, I:].clone().requires_grad_(True)
    (F.silu(gg) * uu).backward(dh)
    out = torch.zeros(R, 2 * I, dtype=torch.bfloat16, device=dev)
    ops.swiglu_bwd(dh.to(dev), gu.to(dev), I, out, total_rows=None, max_rows=R)
    assert rel(out[:, :I].cpu(), gg.grad) < 2 ** -7 and rel(out[:, I:].cpu(), uu.grad) < 2 ** -7


def test_combine_and_permute_bwd_vs_autograd(dev):
    from unimoe_audio_amd import ops
    g = torch.Generator().manual_seed(3)
    S, D, n_real, n_dyn, n_fix = 40, 64, 4, 5, 2
    mask = (torch.rand(S, n_dyn + n_fix, generator=g) < 0.5).to(torch.int32)
    d = ops.dispatch_build_aligned(mask.to(dev), n_real, 8)
    slot_of = d["slot_of"].cpu().long()
    cap = (d["cap"] + 7) // 8 * 8
    y = torch.randn(cap + n_fix * S, D, generator=g).to(torch.bfloat16)
    w = torch.rand(S, n_real, generator=g)
    gw = torch.rand(S, n_dyn + n_fix, generator=g)
    dout = torch.randn(S, D, generator=g).to(torch.bfloat16)
    # reference: out[s] = sum_e w[s,e] y[slot] + sum_i gw[s, n_dyn+i] ysh[i][s]
    yr, wr, gr = y.float().clone().requires_grad_(True), w.clone().requires_grad_(True), gw.clone().requires_grad_(True)
    out = torch.zeros(S, D)
    for e in range(n_real):
        sel = slot_of[:, e] >= 0
        out[sel] = out[sel] + wr[sel, e:e + 1] * yr[slot_of[sel, e]]
    for i in range(n_fix):
        out = out + gr[:, n_dyn + i:n_dyn + i + 1] * yr[cap + i * S: cap + (i + 1) * S]
    out.backward(dout.float())
    yd = y.to(dev)
    dy = torch.zeros_like(yd)
    d_mw, d_gs = ops.combine_bwd(dout.to(dev), yd, d["slot_of"], w.to(dev), yd[cap:], gw.to(dev), n_dyn, n_fix, dy, dy[cap:])
    sel = slot_of >= 0
    assert rel(d_mw.cpu()[sel], wr.grad[sel]) < 1e-3 and float(d_mw.cpu()[~sel].abs().sum()) == 0.0
    assert rel(d_gs.cpu(), gr.grad[:, n_dyn:]) < 1e-3
    used = torch.zeros(cap + n_fix * S, dtype=torch.bool)
    used[slot_of[sel]] = True
    used[cap:] = True
    assert rel(dy.cpu()[used], yr.grad[used]) < 2 ** -7
    # permute backward: dx[s] = sum of the token's slot rows + shared rows + extra
    dxe = torch.randn(cap + n_fix * S, D, generator=g).to(torch.bfloat16)
    extra = torch.randn(S, D, generator=g).to(torch.bfloat16)
    ref = extra.float().clone()
    for e in range(n_real):
        s_ = slot_of[:, e] >= 0
        ref[s_] += dxe.float()[slot_of[s_, e]]
    for i in range(n_fix):
        ref += dxe.float()[cap + i * S: cap + (i + 1) * S]
    dx = ops.permute_bwd(dxe.to(dev), d["slot_of"], dxe.to(dev)[cap:], n_fix, extra=extra.to(dev)).cpu()
    assert rel(dx, ref) < 2 ** -7


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_router_and_aux_bwd_vs_autograd_oracle(dev, dt):
    """d(moe_w), d(shared weights), d(aux) -> d(logits): HIP kernels against torch autograd through the differentiable
    routing restatement (oracle/dcmoe_autograd.py) on the SAME logits and the SAME integer decisions."""
    from unimoe_audio_amd import ops
    from oracle import dcmoe_autograd as OA
    g = torch.Generator().manual_seed(4)
    S, n_dyn, n_real, n_fix, eps = 300, 9, 8, 2, 0.01
    E = n_dyn + n_fix
    logits = (torch.randn(S, E, generator=g) * 1.2).to(dt)
    am = torch.ones(S, dtype=torch.bool)
    am[:17] = False
    r = ops.router_fwd(None, None, n_dyn=n_dyn, n_real=n_real, n_fix=n_fix, top_p=0.7, jitter_eps=eps, logits_in=logits.to(dev),
                       attn_mask=am.to(dev))
    mask, k, sel = r["expert_mask"].cpu(), r["top_k"].cpu(), r["sel"].cpu()
    z = logits.float().clone().requires_grad_(True)                      # fp32 autograd on the same values
    rw, picked = OA.routing_weights(z[:, :n_dyn], k, eps, forced_set=None)
    assert torch.equal(picked * am[:, None].int(), mask[:, :n_dyn])      # same integer decisions
    rw = rw / (rw.sum(-1, keepdim=True) + 1e-6)
    G = torch.softmax(z.masked_fill(mask == 0, float("-inf")), dim=-1)
    gw = torch.cat([rw * G[:, :n_dyn].sum(-1, keepdim=True), G[:, n_dyn:]], -1)
    moe_w = gw[:, :n_real] * mask[:, :n_real]
    d_mw = torch.randn(S, n_real, generator=g)
    d_gs = torch.randn(S, n_fix, generator=g)
    tokw = torch.rand(S, generator=g)
    aux = OA.aux_loss(mask, n_dyn, z, tokw.reshape(1, S))
    ((moe_w * d_mw).sum() + (gw[:, n_dyn:] * d_gs).sum() + 0.7 * aux).backward()
    d_aux = ops.aux_loss_bwd(logits.to(dev), mask.to(dev), n_dyn, tokw.to(dev), torch.tensor(0.7, device=dev))
    got = ops.router_bwd(logits.to(dev), r["sel"], r["top_k"], r["expert_mask"], d_mw.to(dev), d_gs.to(dev), d_aux, n_dyn, n_real,
                         n_fix, eps).cpu()
    assert torch.allclose(r["moe_weight"].cpu(), moe_w.detach(), rtol=2 ** -6 if dt == torch.bfloat16 else 1e-4, atol=1e-4)
    assert rel(got, z.grad) < 2e-3, rel(got, z.grad)


def test_rmsnorm_bwd_vs_autograd(dev):
    from unimoe_audio_amd import ops
    g = torch.Generator().manual_seed(5)
    S, D, eps = 700, 256, 1e-6
    x = torch.randn(S, D, generator=g).to(torch.bfloat16)
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(torch.bfloat16)
    dy = torch.randn(S, D, generator=g).to(torch.bfloat16)
    xr, wr = x.float().clone().requires_grad_(True), w.float().clone().requires_grad_(True)
    (wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + eps))).backward(dy.float())
    dx, dw = ops.rmsnorm_bwd(x.to(dev), w.to(dev), dy.to(dev), eps)
    assert rel(dx.cpu(), xr.grad) < 2 ** -7 and rel(dw.cpu(), wr.grad) < 2 ** -6


@pytest.mark.parametrize("B,T,H,KVH", [(2, 45, 4, 2), (2, 333, 16, 2), (1, 1100, 8, 1), (2, 9, 4, 2)])
def test_attention_backward_composite_vs_autograd(dev, B, T, H, KVH):
    """umoe_attn_prefill_bwd + umoe_qkv_mrope_bwd (through RopeAttentionFn) against autograd of the attention oracle
    (oracle/decode.py: mRoPE, causal GQA softmax in fp32, left padding).  T >= 16: the fused flash-style kernels (dQ with the
    query tile stationary, dK / dV with the key tile stationary); T < 16: the unfused composite on the tiled GEMM."""
    from unimoe_audio_amd import ops, train as TR
    from oracle import decode as OD
    g = torch.Generator().manual_seed(6 + T)
    hd = 128
    D = H * hd
    sections = [16, 24, 24]
    cfg = types.SimpleNamespace(num_attention_heads=H, num_key_value_heads=KVH, hidden_size=D, mrope_section=sections)
    qkv = (torch.randn(B * T, (H + 2 * KVH) * hd, generator=g) * 0.7).to(torch.bfloat16)
    G_ = torch.randn(B * T, D, generator=g).to(torch.bfloat16)
    am = torch.ones(B, T, dtype=torch.long)
    am[0, :min(7, T // 3)] = 0
    pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
    # oracle: identity projections so that x -> (q, k, v) are the given tensors
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
    # HIP
    xg = qkv.to(dev).requires_grad_(True)
    cosd, sind = ops.rope_tables(int(pos.max()) + 2, hd, 1000000.0, dev)
    pos3 = pos[None].expand(3, -1, -1).reshape(3, B * T).to(torch.int32).contiguous().to(dev)
    kv_pos = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
    fv = (am != 0).float().argmax(-1).to(torch.int32)
    ao = TR.RopeAttentionFn.apply(xg, cosd, sind, pos3, kv_pos, fv.to(dev), fv.tolist(), B, T, H, KVH, hd, tuple(sections))
    (ao.float() * (G_.to(dev).float() * valid.to(dev)[:, None])).sum().backward()
    assert rel(ao.detach().cpu()[valid], o.detach()[valid]) < 2.5 * 2 ** -8
    assert rel(xg.grad.cpu()[valid], x.grad[valid]) < 0.02, rel(xg.grad.cpu()[valid], x.grad[valid])


def test_linear_fn_backward(dev):
    from unimoe_audio_amd import train as TR
    g = torch.Generator().manual_seed(7)
    S, K, N = 333, 264, 1027                      # N not a multiple of 8: the padded-operand path
    x = torch.randn(S, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g).to(torch.bfloat16)
    dy = torch.randn(S, N, generator=g).to(torch.bfloat16)
    xr, wr, br = [t.float().clone().requires_grad_(True) for t in (x, w, b)]
    F.linear(xr, wr, br).backward(dy.float())
    xg, wg, bg = [t.to(dev).requires_grad_(True) for t in (x, w, b)]
    TR.LinearFn.apply(xg, wg, bg).backward(dy.to(dev))
    assert rel(xg.grad.cpu(), xr.grad) < 2 ** -7 and rel(wg.grad.cpu(), wr.grad) < 2 ** -7 and rel(bg.grad.cpu(), br.grad) < 2 ** -7
