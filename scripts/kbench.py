"""Kernel micro-benchmarks at the decode shapes (16 rows, full utils/config.json sizes).  Weights rotate over several
distinct copies so every launch streams cold from HBM (the Infinity Cache holds 256 MiB).  Not a test."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimoe_audio_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
D, Id, Is, S = 2048, 2752, 1376, 16


def timeit(fn, iters=60, warm=5):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.02).to(torch.bfloat16)


res = {}
which = sys.argv[1:] or ["dense", "experts", "router", "attn", "sample"]
x = rnd(S, D) * 50

if "dense" in which:
    for name, N, K, norm, resid, f32 in [("qkv", 2560, 2048, True, False, False), ("oproj", 2048, 2048, False, True, False),
                                          ("head", 12324, 2048, True, False, True)]:
        R = 12
        ws = [ops.pack_weight(rnd(N, K)) for _ in range(R)]
        nw = torch.ones(K, device=dev, dtype=torch.bfloat16)
        r = rnd(S, N)
        xin = rnd(S, K)
        tabs = [ops.GroupTable([dict(w=w, static_count=S, n_blocks=(N + 15) // 16, k=K)], dev) for w in ws]
        out = torch.empty((S, N), dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
        epi = ops.EPI_F32 if f32 else (ops.EPI_BF16_RESID if resid else ops.EPI_BF16)
        for nt in (1, 2, 4, 8):
            t = timeit(lambda i: ops.grouped_gemm(tabs[i % R], xin, out, max_rows=S, prologue=ops.PRO_RMSNORM if norm else ops.PRO_PLAIN,
                                                  epilogue=epi, norm_w=nw if norm else None, resid=r if resid else None, n_valid=N, nt=nt))
            res[f"{name}_nt{nt}"] = round(t, 2)
            print(name, "nt", nt, f"{t:.2f} us  {N*K*2/t/1e3:.0f} GB/s", flush=True)

if "experts" in which:
    R = 4
    mask = (torch.rand(S, 11, device=dev) < 0.45).to(torch.int32)
    mask[:, 9:] = 1
    disp = ops.dispatch_build(mask, 8)
    slots = S * 8
    sets = []
    for _ in range(R):
        gu = [ops.pack_gate_up(rnd(Id, D), rnd(Id, D)) for _ in range(8)] + [ops.pack_gate_up(rnd(Is, D), rnd(Is, D)) for _ in range(2)]
        dn = [ops.pack_weight(rnd(D, Id)) for _ in range(8)] + [ops.pack_weight(rnd(D, Is)) for _ in range(2)]
        g1, g2 = [], []
        for e in range(8):
            off, cnt = disp["offsets"][e:e + 1], disp["counts"][e:e + 1]
            g1.append(dict(w=gu[e], rows=disp["slot_token"], row_off=off, count=cnt, n_blocks=2 * Id // 16, k=D))
            g2.append(dict(w=dn[e], row_off=off, count=cnt, n_blocks=D // 16, k=Id))
        for i in range(2):
            g1.append(dict(w=gu[8 + i], static_count=S, out_row_base=slots + i * S, n_blocks=2 * Is // 16, k=D))
            g2.append(dict(w=dn[8 + i], static_count=S, a_row_base=slots + i * S, out_row_base=slots + i * S, n_blocks=D // 16, k=Is))
        sets.append((ops.GroupTable(g1, dev), ops.GroupTable(g2, dev)))
    hbuf = torch.zeros(slots + 2 * S, Id, device=dev, dtype=torch.bfloat16)
    ybuf = torch.zeros(slots + 2 * S, D, device=dev, dtype=torch.bfloat16)
    gub = (8 * 2 * Id * D + 2 * 2 * Is * D) * 2
    dnb = (8 * Id * D + 2 * Is * D) * 2
    for nt in (2, 4, 6, 8):
        t = timeit(lambda i: ops.grouped_gemm(sets[i % R][0], x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, n_valid=Id, nt=nt))
        res[f"gateup_nt{nt}"] = round(t, 2)
        print("gateup nt", nt, f"{t:.2f} us  {gub/t/1e3:.0f} GB/s", flush=True)
    for nt in (1, 2, 4, 8):
        t = timeit(lambda i: ops.grouped_gemm(sets[i % R][1], hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16, n_valid=D, nt=nt))
        res[f"down_nt{nt}"] = round(t, 2)
        print("down nt", nt, f"{t:.2f} us  {dnb/t/1e3:.0f} GB/s", flush=True)
    for nt, wv in ((8, 8), (5, 4), (5, 8), (6, 4), (6, 8), (4, 8)):
        t = timeit(lambda i: ops.grouped_gemm(sets[i % R][1], hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16, n_valid=D, nt=nt, waves=wv))
        print("down nt", nt, "waves", wv, f"{t:.2f} us  {dnb/t/1e3:.0f} GB/s", flush=True)
    t = timeit(lambda i: ops.grouped_gemm(sets[i % R][0], x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, n_valid=Id, nt=8, waves=8))
    print("gateup nt 8 waves 8", f"{t:.2f} us  {gub/t/1e3:.0f} GB/s", flush=True)

if "mall" in which:
    # Does the Infinity Cache (256 MiB) feed a weight-streaming kernel faster than HBM?  One contiguous weight buffer per
    # set so that a prefetch kernel can touch exactly the bytes the GEMM streams next.
    R = 4
    mask = torch.ones(S, 11, device=dev, dtype=torch.int32)
    slots = S * 8
    def make_set():
        per_gu, per_sh = 2 * Id * D, 2 * Is * D
        buf = torch.empty(8 * per_gu + 2 * per_sh, device=dev, dtype=torch.bfloat16)
        g1 = []
        for e in range(8):
            w = buf[e * per_gu:(e + 1) * per_gu]
            w.copy_(ops.pack_gate_up(rnd(Id, D), rnd(Id, D)).view(-1))
            g1.append(dict(w=w, static_count=S, out_row_base=e * S, n_blocks=2 * Id // 16, k=D))
        for i in range(2):
            w = buf[8 * per_gu + i * per_sh: 8 * per_gu + (i + 1) * per_sh]
            w.copy_(ops.pack_gate_up(rnd(Is, D), rnd(Is, D)).view(-1))
            g1.append(dict(w=w, static_count=S, out_row_base=slots + i * S, n_blocks=2 * Is // 16, k=D))
        return buf, ops.GroupTable(g1, dev)
    sets = [make_set() for _ in range(R)]
    hbuf = torch.zeros(slots + 2 * S, Id, device=dev, dtype=torch.bfloat16)
    nbytes = sets[0][0].numel() * 2
    def gemm(i, pol): ops.grouped_gemm(sets[i][1], x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, n_valid=Id, nt=8, cache_policy=pol)
    for pol in (0, 1):
        t = timeit(lambda i: gemm(i % R, pol))
        print(f"gate_up dense policy {pol} rotating {R} sets (cold): {t:.2f} us  {nbytes/t/1e3:.0f} GB/s", flush=True)
        t = timeit(lambda i: gemm(0, pol))
        print(f"gate_up dense policy {pol} same set (warm): {t:.2f} us  {nbytes/t/1e3:.0f} GB/s", flush=True)
    for wgs in (256, 512, 1024, 2048):
        t = timeit(lambda i: ops.prefetch(sets[i % R][0], wgs))
        print(f"prefetch {nbytes/1e6:.0f} MB with {wgs} workgroups (cold): {t:.2f} us  {nbytes/t/1e3:.0f} GB/s", flush=True)
    for frac in (0.25, 0.5, 1.0):
        nb = int(nbytes * frac) // 4096 * 4096
        for pol in (0, 1):
            def both(i):
                ops.prefetch(sets[i % R][0], 1024, nb)
                gemm(i % R, pol)
            tb = timeit(both)
            tp = timeit(lambda i: ops.prefetch(sets[i % R][0], 1024, nb))
            print(f"prefetch {frac:.2f} of the set then gate_up policy {pol}: pair {tb:.2f} us, prefetch alone {tp:.2f} us -> gate_up {tb - tp:.2f} us", flush=True)

if "tiled" in which:
    # compute-bound shapes: prefill (4800 tokens) and training (6240 tokens) through umoe_tiled_gemm
    for M in (4800, 6240):
        xa = rnd(M, D) * 50
        for name, N, K in (("qkv", 2560, D), ("o_proj", D, D), ("head", 12324, D)):
            w = rnd(N, K)
            xin = rnd(M, K)
            t = timeit(lambda i: ops.tlinear(xin, w), iters=20)
            print(f"tiled {name} M={M} N={N} K={K}: {t:.1f} us  {2*M*N*K/t/1e6:.0f} TFLOP/s", flush=True)
        wo_, xo_, ro_ = rnd(D, D), rnd(M, D), rnd(M, D)
        t = timeit(lambda i: ops.tlinear(xo_, wo_, resid=ro_), iters=20)
        print(f"tiled o_proj+resid M={M}: {t:.1f} us  {2*M*D*D/t/1e6:.0f} TFLOP/s", flush=True)
        wg, wu, wd = rnd(Id, D), rnd(Id, D), rnd(D, Id)
        hb = torch.empty(M, Id, device=dev, dtype=torch.bfloat16)
        yb = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
        t = timeit(lambda i: ops.tiled_gemm([dict(w=wg, w2=wu, static_count=M)], xa, hb, max_rows=M, epilogue=ops.EPI_SWIGLU), iters=20)
        print(f"tiled gate/up SwiGLU M={M}: {t:.1f} us  {4*M*Id*D/t/1e6:.0f} TFLOP/s", flush=True)
        t = timeit(lambda i: ops.tiled_gemm([dict(w=wd, static_count=M)], hb, yb, max_rows=M, epilogue=ops.EPI_BF16), iters=20)
        print(f"tiled down M={M}: {t:.1f} us  {2*M*Id*D/t/1e6:.0f} TFLOP/s", flush=True)
        xin = rnd(M, D)
        w = rnd(2560, D)
        t = timeit(lambda i: torch.nn.functional.linear(xin, w), iters=20)
        print(f"torch (hipBLASLt) qkv M={M}: {t:.1f} us  {2*M*2560*D/t/1e6:.0f} TFLOP/s", flush=True)
        wp = ops.pack_weight(w)
        t = timeit(lambda i: ops.linear(xin, wp, 2560), iters=5)
        print(f"weight-streaming qkv M={M}: {t:.1f} us  {2*M*2560*D/t/1e6:.0f} TFLOP/s", flush=True)

if "train" in which:
    # BASELINE configs[2] shape for ONE DCMoE block: 6240 tokens, full utils/config.json sizes, forward + backward
    from unimoe_audio_amd.config import UniMoEAudioConfig
    from unimoe_audio_amd.dcmoe import UniMoEAudioSparseMoeBlock
    import time
    cfg = UniMoEAudioConfig()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        blk = UniMoEAudioSparseMoeBlock(cfg)
    torch.set_default_dtype(torch.float32)
    with torch.no_grad():
        for n, p_ in blk.named_parameters():
            p_.normal_(0, 0.02)
    blk.train(True)
    for M in (1560, 6240):
        xt = (torch.randn(1, M, D, device=dev) * 1.0).to(torch.bfloat16).requires_grad_(True)
        Gt = torch.randn(1, M, D, device=dev).to(torch.bfloat16)
        def step():
            for p_ in blk.parameters():
                p_.grad = None
            o = blk(xt, None, None)
            ((o[0].float() * Gt.float()).sum() + 0.01 * o[5].float()).backward()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_it = 5
        for _ in range(n_it):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_it
        with torch.no_grad():
            o = blk(xt.detach(), None, None)
            kreal = float(o[3][:, :8].sum()) / M
        flops = 3 * 2 * M * (kreal * 3 * Id * D + 2 * 3 * Is * D)       # fwd + 2x bwd, routed k_real + 2 shared experts
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_it):
            with torch.no_grad():
                blk(xt.detach(), None, None)
        torch.cuda.synchronize()
        dtf = (time.perf_counter() - t0) / n_it
        print(f"DCMoE block M={M}: fwd+bwd {dt*1e3:.2f} ms ({flops/dt/1e12:.0f} TFLOP/s algorithmic, k_real={kreal:.2f}), fwd only {dtf*1e3:.2f} ms", flush=True)

if "dense" in which and "gateup" in which:
    pass
if "flat" in which:
    # dense-expert decode layout (every expert computes all 16 rows): workgroup count vs CUs
    gu = [ops.pack_gate_up(rnd(Id, D), rnd(Id, D)) for _ in range(8)] + [ops.pack_gate_up(rnd(Is, D), rnd(Is, D)) for _ in range(2)]
    dn = [ops.pack_weight(rnd(D, Id)) for _ in range(8)] + [ops.pack_weight(rnd(D, Is)) for _ in range(2)]
    R = 4
    sets = []
    for r in range(R):
        gu_r = [g.clone() for g in gu]
        dn_r = [g.clone() for g in dn]
        g1 = [dict(w=gu_r[e], static_count=S, out_row_base=e * S, n_blocks=2 * (Id if e < 8 else Is) // 16, k=D) for e in range(10)]
        g2 = [dict(w=dn_r[e], static_count=S, a_row_base=e * S, out_row_base=e * S, n_blocks=D // 16, k=(Id if e < 8 else Is)) for e in range(10)]
        sets.append((ops.GroupTable(g1, dev), ops.GroupTable(g2, dev)))
    hbuf = torch.zeros(10 * S, Id, device=dev, dtype=torch.bfloat16)
    ybuf = torch.zeros(10 * S, D, device=dev, dtype=torch.bfloat16)
    gub = (8 * 2 * Id * D + 2 * 2 * Is * D) * 2
    dnb = (8 * Id * D + 2 * Is * D) * 2
    for nt in (8, 14):
        t = timeit(lambda i: ops.grouped_gemm(sets[i % R][0], x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, n_valid=Id, nt=nt))
        print("dense gateup nt", nt, f"{t:.2f} us  {gub/t/1e3:.0f} GB/s", flush=True)
    for nt, wv in ((8, 8), (6, 8), (5, 8)):
        t = timeit(lambda i: ops.grouped_gemm(sets[i % R][1], hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16, n_valid=D, nt=nt, waves=wv))
        print("dense down nt", nt, "waves", wv, f"{t:.2f} us  {dnb/t/1e3:.0f} GB/s", flush=True)

    # NT 5 gives 260 workgroups (4 more than CUs): does the ORDER of the groups decide which CUs carry two?
    for name, order in (("shared last", list(range(10))), ("shared first+last", [8] + list(range(8)) + [9]), ("shared first", [8, 9] + list(range(8)))):
        tabs = []
        for r in range(R):
            dn_r = [g.clone() for g in dn]
            g2 = [dict(w=dn_r[e], static_count=S, a_row_base=e * S, out_row_base=e * S, n_blocks=D // 16, k=(Id if e < 8 else Is)) for e in order]
            tabs.append(ops.GroupTable(g2, dev))
        for nt in (5, 6):
            t = timeit(lambda i: ops.grouped_gemm(tabs[i % R], hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16, n_valid=D, nt=nt, waves=8))
            print("dense down order", name, "nt", nt, f"{t:.2f} us  {dnb/t/1e3:.0f} GB/s", flush=True)

if "ragged" in which:
    # in-situ shape of the routed experts' gate/up at training size: 6240 tokens, ~3.55 of 8 experts per token
    M = 6240
    xa = rnd(M, D) * 50
    mask = (torch.rand(M, 11, device=dev) < 3.55 / 8).to(torch.int32)
    disp = ops.dispatch_build(mask, 8)
    slots = int(disp["offsets"][8].item())
    wgs, wus = [rnd(Id, D) for _ in range(8)], [rnd(Id, D) for _ in range(8)]
    hb = torch.empty(slots + 64, Id, device=dev, dtype=torch.bfloat16)
    aux = torch.empty(slots + 64, 2 * Id, device=dev, dtype=torch.bfloat16)
    fl = 4 * slots * Id * D
    g_r = [dict(w=wgs[e], w2=wus[e], rows=disp["slot_token"], row_off=disp["offsets"][e:e + 1], count=disp["counts"][e:e + 1]) for e in range(8)]
    t = timeit(lambda i: ops.tiled_gemm(g_r, xa, hb, max_rows=M, epilogue=ops.EPI_SWIGLU), iters=20)
    print(f"ragged gate/up 8 experts, {slots} slots, gather list: {t:.1f} us  {fl/t/1e6:.0f} TFLOP/s", flush=True)
    t = timeit(lambda i: ops.tiled_gemm(g_r, xa, hb, max_rows=M, epilogue=ops.EPI_SWIGLU, aux_out=aux), iters=20)
    print(f"  + pre-activation outputs (training forward): {t:.1f} us  {fl/t/1e6:.0f} TFLOP/s", flush=True)
    sets = []
    for r in range(3):       # 3 x 180 MB of expert weights: more than the 256 MiB Infinity Cache -> every launch streams from HBM
        wg2, wu2 = [rnd(Id, D) for _ in range(8)], [rnd(Id, D) for _ in range(8)]
        sets.append([dict(w=wg2[e], w2=wu2[e], rows=disp["slot_token"], row_off=disp["offsets"][e:e + 1], count=disp["counts"][e:e + 1]) for e in range(8)])
    t = timeit(lambda i: ops.tiled_gemm(sets[i % 3], xa, hb, max_rows=M, epilogue=ops.EPI_SWIGLU, aux_out=aux), iters=21)
    print(f"  + weights rotating over 3 sets (cold, as inside the layer loop): {t:.1f} us  {fl/t/1e6:.0f} TFLOP/s", flush=True)
    t = timeit(lambda i: ops.tiled_gemm(sets[i % 3], xa, hb, max_rows=M, epilogue=ops.EPI_SWIGLU, aux_out=aux), iters=1500, warm=300)
    print(f"  the same, 1500 launches back to back (~1 s of sustained MFMA load: clocks): {t:.1f} us  {fl/t/1e6:.0f} TFLOP/s", flush=True)
    xs = torch.empty(slots + 64, D, device=dev, dtype=torch.bfloat16).normal_()
    g_c = [dict(w=wgs[e], w2=wus[e], row_off=disp["offsets"][e:e + 1], count=disp["counts"][e:e + 1]) for e in range(8)]
    t = timeit(lambda i: ops.tiled_gemm(g_c, xs, hb, max_rows=M, epilogue=ops.EPI_SWIGLU), iters=20)
    print(f"  contiguous slot rows instead of the gather list: {t:.1f} us  {fl/t/1e6:.0f} TFLOP/s", flush=True)
    per = slots // 8
    g_s = [dict(w=wgs[e], w2=wus[e], static_count=per, a_row_base=e * per, out_row_base=e * per) for e in range(8)]
    t = timeit(lambda i: ops.tiled_gemm(g_s, xs, hb, max_rows=per, epilogue=ops.EPI_SWIGLU), iters=20)
    print(f"  static groups of {per} rows (tight grid, contiguous XCD ranges): {t:.1f} us  {4*8*per*Id*D/t/1e6:.0f} TFLOP/s", flush=True)

if "router" in which:
    gw = rnd(11, D)
    nw = torch.ones(D, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda i: ops.router_fwd(x, gw, n_dyn=9, n_real=8, n_fix=2, top_p=0.7, norm_w=nw, want_h=True))
    res["router"] = round(t, 2)
    print("router+alloc", f"{t:.2f} us", flush=True)
    t = timeit(lambda i: ops.router_dispatch_fwd(x, gw, n_dyn=9, n_real=8, n_fix=2, top_p=0.7, norm_w=nw))
    print("router_dispatch fused + alloc", f"{t:.2f} us", flush=True)
    lg = (torch.randn(S, 11, device=dev) * 0.9).to(torch.bfloat16)
    t = timeit(lambda i: ops.router_fwd(None, None, n_dyn=9, n_real=8, n_fix=2, top_p=0.7, logits_in=lg))
    print("router logits-only + alloc", f"{t:.2f} us", flush=True)
    e = torch.empty(16, device=dev)
    t = timeit(lambda i: [torch.empty((S, 11), device=dev) for _ in range(8)])
    print("8 torch.empty allocs", f"{t:.2f} us", flush=True)
    gws = [rnd(11, D) for _ in range(64)]
    t = timeit(lambda i: ops.router_fwd(x, gws[i % 64], n_dyn=9, n_real=8, n_fix=2, top_p=0.7, norm_w=nw, want_h=True))
    print("router cold gate_w + alloc", f"{t:.2f} us", flush=True)

if "attn" in which:
    rows, KVH, H, hd, Lmax, L = 16, 2, 16, 128, 1024, 550
    kc, vc = rnd(rows, KVH, Lmax, hd) * 50, rnd(rows, KVH, Lmax, hd) * 50
    q = rnd(rows, H * hd) * 50
    ks = torch.zeros(rows, dtype=torch.int32, device=dev)
    q0 = torch.full((rows,), L, dtype=torch.int32, device=dev)
    for splits in (1, 2, 4, 8, 16):
        t = timeit(lambda i: ops.attention(q, kc, vc, ks, q0, 1, H, splits=splits))
        res[f"attn_s{splits}"] = round(t, 2)
        print("attn splits", splits, f"{t:.2f} us (2 launches + 3 allocs)", flush=True)

if "sample" in which:
    lg = torch.randn(16, 12 * 1027, device=dev)
    t = timeit(lambda i: ops.cfg_sample(lg, 8, 12, 1027, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos=1024, eos_mul=0.8, seed=i), iters=20)
    res["sample"] = round(t, 2)
    print("sample", f"{t:.2f} us", flush=True)
    t = timeit(lambda i: ops.cfg_sample(lg, 8, 12, 1027, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos=1024, eos_mul=0.8, do_sample=False), iters=20)
    print("argmax", f"{t:.2f} us", flush=True)
print(json.dumps(res))
