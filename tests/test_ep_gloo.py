"""CPU suite: the expert-parallel exchange (unimoe_audio_amd/ep.py) with world_size 2 over gloo.  The expert function
is the CPU oracle's SwiGLU; the result on every rank must equal the single-process oracle DCMoE on that rank's tokens.
Also covers bench.py's multi-rank aggregation (max over ranks)."""
import os
import types

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _cfg():
    return types.SimpleNamespace(hidden_size=64, mlp_dynamic_expert_num=8, mlp_dynamic_null_expert_num=1, mlp_dynamic_top_p=0.7,
                                 mlp_dynamic_top_k=2, mlp_fixed_expert_num=2, router_jitter_noise=0.01, token_drop=False,
                                 fp32_gate=True, capacity_factor=6.0, min_capacity=8, drop_policy="probs",
                                 dynamic_intermediate_size=96, shared_intermediate_size=64)


def _weights(cfg, seed):
    from oracle.dcmoe import EXPERT_FMT, SHARED_FMT
    g = torch.Generator().manual_seed(seed)
    w = {"gate.weight": (torch.randn(11, 64, generator=g) * 0.3).to(torch.bfloat16)}
    for e in range(8):
        for p, shp in (("gate", (96, 64)), ("up", (96, 64)), ("down", (64, 96))):
            w[EXPERT_FMT.format(e=e, p=p)] = (torch.randn(*shp, generator=g) * 0.08).to(torch.bfloat16)
    for i in range(2):
        for p, shp in (("gate", (64, 64)), ("up", (64, 64)), ("down", (64, 64))):
            w[SHARED_FMT.format(i=i, p=p)] = (torch.randn(*shp, generator=g) * 0.08).to(torch.bfloat16)
    return w


def _worker(rank, world, port, S, out_q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import router as OR
        from oracle.dcmoe import DCMoEOracle, EXPERT_FMT, swiglu_mlp
        from unimoe_audio_amd import ep as EP
        cfg, w = _cfg(), _weights(_cfg(), 5)
        n_real, n_dyn, n_fix = 8, 9, 2
        E_loc = n_real // world
        torch.manual_seed(100 + rank)
        x = torch.randn(1, S, 64).to(torch.bfloat16)
        ref = DCMoEOracle(cfg, w)(x, None, None)
        h = x[0]
        logits = torch.nn.functional.linear(h, w["gate.weight"])
        r = OR.route(logits, n_dyn, n_real, n_fix, 0.7, 0, 0.01, None)
        d = OR.dispatch(r["expert_mask"], n_real)
        disp = dict(counts=d["counts"], offsets=d["offsets"], slot_token=torch.cat([d["slot_token"], torch.zeros(1, dtype=torch.int32)]),
                    slot_of=d["slot_of"])

        def expert_fn(recv, cnt):            # recv [ep, S, E_loc, D]: my local experts on the rows of every source rank
            y = torch.zeros_like(recv)
            for e_loc in range(E_loc):
                e = rank * E_loc + e_loc
                for src in range(world):
                    n = int(cnt[src, e_loc])
                    if n:
                        y[src, :n, e_loc] = swiglu_mlp(recv[src, :n, e_loc], w[EXPERT_FMT.format(e=e, p="gate")],
                                                       w[EXPERT_FMT.format(e=e, p="up")], w[EXPERT_FMT.format(e=e, p="down")])
            return y
        y_back, slot_of_ep = EP.ep_moe(h, disp, n_real, world, dist.group.WORLD, expert_fn)
        # combine exactly as the single-GPU path does (ascending expert order, fp32 accumulate, one rounding)
        moe_w = r["moe_weight"].float()
        acc = torch.zeros(S, 64)
        for e in range(n_real):
            so = slot_of_ep[:, e].long()
            sel = so >= 0
            acc[sel] += moe_w[sel, e:e + 1] * y_back[so[sel]].float()
        moe_out = acc.to(torch.bfloat16)
        # reference MoE-only output = block output minus shared experts: recompute the same way on one process
        one = DCMoEOracle(cfg, w)
        ref_moe = one.moe_layer(h, r["expert_mask"][:, :n_real], r["global_weight"][:, :n_real])
        ok = torch.allclose(moe_out.float(), ref_moe.float(), rtol=2 ** -6, atol=2 ** -9)
        out_q.put((rank, bool(ok), float((moe_out.float() - ref_moe.float()).abs().max())))
        # the RAGGED exchange (counts in a header, per-destination row ranges; what the HIP block runs): the outputs come back in this rank's
        # own slot order, so the combine uses the LOCAL slot_of table; also with 8-aligned expert blocks (the training path's dispatch)
        for align in (1, 8):
            offs_a, run = [], 0
            for e in range(n_real):
                offs_a.append(run)
                run = (run + int(d["counts"][e]) + align - 1) // align * align
            offs_a.append(run)
            offs_a = torch.tensor(offs_a, dtype=torch.int32)
            st_a = torch.zeros(max(run, 1), dtype=torch.int32)
            so_a = torch.full((S, n_real), -1, dtype=torch.int32)
            for e in range(n_real):
                toks = (r["expert_mask"][:, e] != 0).nonzero().flatten()
                st_a[int(offs_a[e]): int(offs_a[e]) + len(toks)] = toks.to(torch.int32)
                so_a[toks, e] = int(offs_a[e]) + torch.arange(len(toks), dtype=torch.int32)
            disp_a = dict(counts=d["counts"], offsets=offs_a, slot_token=st_a, slot_of=so_a)

            def expert_fn_r(recv, plan):
                y2 = torch.zeros((plan.cap2, recv.shape[1]), dtype=recv.dtype)
                for e_loc in range(E_loc):
                    e = rank * E_loc + e_loc
                    o, n = int(plan.offsets2[e_loc]), int(plan.counts2[e_loc])
                    assert o % align == 0
                    if n:
                        rows = recv[plan.list2[o: o + n].long()]
                        y2[o: o + n] = swiglu_mlp(rows, w[EXPERT_FMT.format(e=e, p="gate")], w[EXPERT_FMT.format(e=e, p="up")],
                                                  w[EXPERT_FMT.format(e=e, p="down")])
                return y2
            y_slots, plan = EP.ep_moe_ragged(h, disp_a, n_real, world, dist.group.WORLD, expert_fn_r, align=align)
            assert y_slots.shape[0] == int(offs_a[n_real]) and plan.n_recv == sum(plan.out_splits)
            acc3 = torch.zeros(S, 64)
            for e in range(n_real):
                so = so_a[:, e].long()
                sel = so >= 0
                acc3[sel] += moe_w[sel, e:e + 1] * y_slots[so[sel]].float()
            ok3 = torch.equal(acc3.to(torch.bfloat16), moe_out)          # the same rows, the same order: bit-identical to the padded exchange
            out_q.put((rank + 20 + (100 if align == 8 else 0), bool(ok3), 0.0))
        # the DENSE exchange of the decode engine (every row visits every expert, the owner selects by its mask):
        # layout contract of csrc/umoe_engine.hip run_moe_ep, restated in ep.dense_ep_moe
        def one_expert(e, xx):
            return swiglu_mlp(xx, w[EXPERT_FMT.format(e=e, p="gate")], w[EXPERT_FMT.format(e=e, p="up")], w[EXPERT_FMT.format(e=e, p="down")])
        y_dense = EP.dense_ep_moe(h, one_expert, rank, world, n_real, dist.group.WORLD)      # [n_real, S, D]
        acc2 = torch.zeros(S, 64)
        for e in range(n_real):
            sel = r["expert_mask"][:, e] != 0
            acc2[sel] += moe_w[sel, e:e + 1] * y_dense[e][sel].float()
        ok2 = torch.allclose(acc2.to(torch.bfloat16).float(), ref_moe.float(), rtol=2 ** -6, atol=2 ** -9)
        # and every (expert, row) product is the single-process product, bit for bit (what makes ep == 1 and ep > 1 identical)
        for e in range(n_real):
            ok2 = ok2 and torch.equal(y_dense[e], one_expert(e, h))
        out_q.put((rank + 10, bool(ok2), 0.0))
        # bench.py aggregation: MAX over ranks of the timed region
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        out_q.put(("max", float(t)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("S", [5, 16])
def test_ep_exchange_world2_gloo(S):
    import random
    port = random.randint(20000, 40000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, S, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(10)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    oks = [r for r in res if r[0] in (0, 1)]
    assert len(oks) == 2 and all(r[1] for r in oks), oks
    ragged = [r for r in res if r[0] in (20, 21, 120, 121)]
    assert len(ragged) == 4 and all(r[1] for r in ragged), ragged
    dense = [r for r in res if r[0] in (10, 11)]
    assert len(dense) == 2 and all(r[1] for r in dense), dense
    assert all(r[1] == 2.0 for r in res if r[0] == "max")


def test_ep_index_math_single_process():
    """ep_size 1 must be the identity exchange (the reference's single-process patch, utils.py:332-335)."""
    from oracle import router as OR
    from unimoe_audio_amd import ep as EP
    torch.manual_seed(3)
    S, D, n_real = 7, 16, 8
    mask = (torch.rand(S, 11) < 0.5).to(torch.int32)
    d = OR.dispatch(mask, n_real)
    h = torch.randn(S, D)
    disp = dict(counts=d["counts"], offsets=d["offsets"], slot_token=torch.cat([d["slot_token"], torch.zeros(1, dtype=torch.int32)]),
                slot_of=d["slot_of"])
    y_back, so = EP.ep_moe(h, disp, n_real, 1, None, lambda recv, cnt: recv * 2.0)
    for s in range(S):
        for e in range(n_real):
            if mask[s, e]:
                assert torch.equal(y_back[int(so[s, e])], h[s] * 2.0)
            else:
                assert int(so[s, e]) == -1


def _grad_worker(rank, world, port, S, out_q):
    """Expert-parallel block under autograd: gradients of the own rows and of the LOCAL experts' weights (which collect the
    contributions of every rank's rows through the backward of the exchange) against single-process autograd over all rows."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import router as OR
        from oracle.dcmoe import EXPERT_FMT
        from unimoe_audio_amd import ep as EP
        w = _weights(_cfg(), 5)
        n_real, n_dyn, n_fix, D = 8, 9, 2, 64
        E_loc = n_real // world
        params = {}
        for e in range(n_real):
            for p in ("gate", "up", "down"):
                params[(e, p)] = w[EXPERT_FMT.format(e=e, p=p)].float().clone().requires_grad_(True)

        def mlp(e, x):      # fp32 SwiGLU (the gradient check is about the exchange, not about bf16 rounding points)
            return torch.nn.functional.linear(torch.nn.functional.silu(torch.nn.functional.linear(x, params[(e, "gate")])) *
                                              torch.nn.functional.linear(x, params[(e, "up")]), params[(e, "down")])
        hs, rs, gs = [], [], []
        for r in range(world):       # every rank builds every rank's inputs (the single-process reference needs them all)
            g = torch.Generator().manual_seed(200 + r)
            hs.append(torch.randn(S, D, generator=g))
            logits = torch.nn.functional.linear(hs[-1].to(torch.bfloat16), w["gate.weight"])
            rs.append(OR.route(logits, n_dyn, n_real, n_fix, 0.7, 0, 0.01, None))
            gs.append(torch.randn(S, D, generator=g))
        h = hs[rank].clone().requires_grad_(True)
        r = rs[rank]
        d = OR.dispatch(r["expert_mask"], n_real)
        disp = dict(counts=d["counts"], offsets=d["offsets"], slot_token=torch.cat([d["slot_token"], torch.zeros(1, dtype=torch.int32)]),
                    slot_of=d["slot_of"])

        def expert_fn(recv, cnt):
            y = torch.zeros_like(recv)
            for e_loc in range(E_loc):
                y[:, :, e_loc] = mlp(rank * E_loc + e_loc, recv[:, :, e_loc])          # zero rows in -> zero rows out (no bias)
            return y
        y_back, so = EP.ep_moe(h, disp, n_real, world, dist.group.WORLD, expert_fn)
        out = EP.ep_combine(y_back, so, r["moe_weight"].float())
        (out * gs[rank]).sum().backward()
        got_dh = h.grad.clone()
        got_dw = {k: v.grad.clone() for k, v in params.items() if k[0] // E_loc == rank}
        for v in params.values():
            v.grad = None
        # single-process reference over ALL ranks' rows
        href = [x.clone().requires_grad_(True) for x in hs]
        tot = 0.0
        for q in range(world):
            mw = rs[q]["moe_weight"].float()
            acc = torch.zeros(S, D)
            for e in range(n_real):
                sel = (rs[q]["expert_mask"][:, e] != 0).float()[:, None]
                acc = acc + sel * mw[:, e:e + 1] * mlp(e, href[q])
            tot = tot + (acc * gs[q]).sum()
        tot.backward()
        ok = torch.allclose(got_dh, href[rank].grad, rtol=1e-4, atol=1e-5)
        err = float((got_dh - href[rank].grad).abs().max())
        for k, v in got_dw.items():
            ok = ok and torch.allclose(v, params[k].grad, rtol=1e-4, atol=1e-5)
            err = max(err, float((v - params[k].grad).abs().max()))
        out_q.put((rank, bool(ok), err))
    finally:
        dist.destroy_process_group()


def test_ep_exchange_backward_world2_gloo():
    """The exchange as an autograd node (ep._AllToAll = DeepSpeed's _AllToAll, core.py:467,480): d(own rows) and d(local experts'
    weights, summed over the rows of BOTH ranks) equal single-process autograd."""
    import random
    port = random.randint(20000, 40000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_grad_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
