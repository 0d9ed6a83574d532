"""Full-depth parity of BASELINE configs[1] (and configs[0]): the 36-layer model at the reference's layer width through the
decode engine -- tiled prefill of a left-padded 64-token prompt, then teacher-forced decode steps, eager AND hipGraph replay --
against the CPU oracle (oracle/decode.py; reference _decoder_step, utils/UniMoE_Audio_model.py:918-1068, layer loop :319-457).
What is measured per step is printed and written to gpurun_out/full_depth_parity.json; the bounds below are those
measurements plus a margin (recorded next to each assert)."""
import json
import os
import time

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

RESULTS = os.path.join(ROOT, "gpurun_out", "full_depth_parity.json")


def _weights_on_device(cfg, dev, seed):
    """N(0, 0.02^2) Linear / Embedding weights, RMSNorm weights 1 + 0.05 n, q/k/v biases 0.02 n: drawn on the device (5.5 B
    parameters), copied once to the host for the oracle."""
    from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration as Model
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device(dev):
            m = Model(cfg)
    finally:
        torch.set_default_dtype(torch.float32)
    g = torch.Generator(device=dev).manual_seed(seed)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "layernorm" in n or n.endswith("norm.weight"):
                p.copy_(1 + 0.05 * torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32))
            elif n.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32))
            else:
                p.copy_(0.02 * torch.randn(p.shape, generator=g, device=dev, dtype=torch.float32))
    m.eval()
    w = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return m, w


@pytest.fixture(scope="module")
def full_model():
    assert torch.cuda.is_available()
    from unimoe_audio_amd.config import UniMoEAudioConfig
    dev = torch.device("cuda:0")
    # the reference architecture (utils/config.json) except the 151 k-entry text vocabulary, which the decode path never touches
    cfg = UniMoEAudioConfig(vocab_size=320, codec_placeholder_value=300)
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.num_key_value_heads) == (2048, 36, 16, 2)
    m, w = _weights_on_device(cfg, dev, 4321)
    yield cfg, m, w, dev
    del m
    torch.cuda.empty_cache()


def _save(key, rec):
    os.makedirs(os.path.dirname(RESULTS), exist_ok=True)
    allr = {}
    if os.path.exists(RESULTS):
        try:
            allr = json.load(open(RESULTS))
        except Exception:
            allr = {}
    allr[key] = rec
    json.dump(allr, open(RESULTS, "w"), indent=1)


@pytest.mark.parametrize("B,ragged", [(8, False), (1, False), (1, True)])
def test_full_depth_teacher_forced_decode_vs_oracle(full_model, B, ragged, monkeypatch):
    """B = 8: 16 CFG rows, dense-expert layout with the router riders and the fused expert launch (BASELINE configs[1]); B = 1: 2 rows
    (configs[0]), in the dense layout (the default from 2 rows) and through the ragged dispatch tables (UMOE_DENSE_MIN_ROWS=6)."""
    monkeypatch.setenv("UMOE_DENSE_MIN_ROWS", "6" if ragged else "2")
    from oracle import decode as OD
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.model import DecodeEngine
    cfg, gm, w, dev = full_model
    T, steps, MAXT = 64, 3, 48
    C, V, E, Lyr = cfg.codec_channels, cfg.codec_vocab_size, cfg.num_experts, cfg.num_hidden_layers
    g = torch.Generator().manual_seed(100 + B)
    ids = torch.randint(0, 290, (2 * B, T), generator=g)
    am = torch.ones(2 * B, T, dtype=torch.long)
    for r in range(0, 2 * B, 2):
        am[r, : 9 + (r % 5)] = 0                       # uncond rows: shorter prompt, left padded (mod.py:456-461)
    n_codec = 24
    ids[:, -n_codec - 3:-3] = cfg.codec_placeholder_value
    codec = torch.randint(0, 1024, (2 * B * n_codec, C), generator=g)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    step0 = min(psteps) - 1
    # the teacher's tokens: random codes wherever the prompt buffer says "to be generated" (-1); its BOS entries (delay
    # pattern, utils.py:137-200) stay -- both sides read the same buffer
    forced = torch.randint(0, 1024, (B, max(pre.shape[1], step0 + steps + 2), C), generator=g).to(torch.int32)
    keep = pre.to(torch.int32) != -1
    forced[:, : pre.shape[1]][keep] = pre.to(torch.int32)[keep]
    # ---- oracle: prefill + teacher-forced steps (CPU)
    t0 = time.time()
    tm = OD.TextModelOracle(cfg, w)
    key_valid = am.bool()
    pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
    x = OD.input_embedding(cfg, w, ids, codec)
    with torch.no_grad():
        _, cache, _ = tm.forward(x, key_valid, pos, None)
        ref = []
        for s in range(steps):
            kv1 = torch.cat([key_valid, torch.ones((2 * B, 1), dtype=torch.bool)], -1)
            p1 = (kv1.long().cumsum(-1) - 1).masked_fill(~kv1, 1)[:, -1:]
            tok2 = forced[:, step0 + s: step0 + s + 1].long().repeat_interleave(2, dim=0)
            h, cache, router = tm.forward(OD.codec_embedding(cfg, w, tok2), kv1, p1, cache, collect_router=True)
            key_valid = kv1
            ref.append((torch.nn.functional.linear(h, w["codec_head.weight"]).float()[:, -1],
                        torch.stack([r["expert_mask"] for r in router]), torch.stack([r["top_k"] for r in router])))
    t_oracle = time.time() - t0
    # ---- the same steps in fp32 (weights upcast, every op in fp32): a neutral centre.  Both bf16 implementations -- the CPU
    # oracle and the HIP path -- differ from it by summation order and bf16 rounding points only, so "parity" at 36 layers is:
    # the HIP path sits no further from the centre than the oracle does (B = 8 only: 22 GB of fp32 weights, ~1 min)
    ref32 = None
    if B == 8:
        w32 = {k: v.float() for k, v in w.items()}
        tm32 = OD.TextModelOracle(cfg, w32)
        kv = am.bool()
        with torch.no_grad():
            _, c32, _ = tm32.forward(OD.input_embedding(cfg, w32, ids, codec), kv, pos, None)
            ref32 = []
            for s in range(steps):
                kv1 = torch.cat([kv, torch.ones((2 * B, 1), dtype=torch.bool)], -1)
                p1 = (kv1.long().cumsum(-1) - 1).masked_fill(~kv1, 1)[:, -1:]
                tok2 = forced[:, step0 + s: step0 + s + 1].long().repeat_interleave(2, dim=0)
                h, c32, router = tm32.forward(OD.codec_embedding(cfg, w32, tok2), kv1, p1, c32, collect_router=True)
                kv = kv1
                ref32.append((torch.nn.functional.linear(h, w32["codec_head.weight"]).float()[:, -1],
                              torch.stack([r["expert_mask"] for r in router])))
        del w32, tm32, c32
    # ---- engine: eager steps, then the same steps again through the captured graph
    xg = gm.calculate_input_embedding(ids.to(dev), codec.to(dev))
    assert torch.equal(xg.cpu(), x)
    runs = {}
    for use_graph in (False, True):
        eng = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64)
        eng.prefill(xg.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev))
        eng.start_decode(forced, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.0, top_p=1.0, top_k=45, eos_mul=0.8, do_sample=False)
        out = []
        for s in range(steps):
            eng.step(use_graph=use_graph)
            out.append((eng.copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu(),
                        eng.copy_buffer("all_mask", torch.int32, (Lyr, 2 * B, E)).cpu(),
                        eng.copy_buffer("all_topk", torch.int64, (Lyr, 2 * B)).cpu(),
                        eng.copy_buffer("pred", torch.int64, (B, C)).cpu()))
        runs[use_graph] = out
        eng.close()
    rec = {"rows": 2 * B, "layers": Lyr, "prompt": T, "steps": steps, "oracle_seconds": round(t_oracle, 1), "per_step": []}
    for s in range(steps):
        got, masks, topk, pred = runs[False][s]
        ref_logits, ref_mask, ref_topk = ref[s]
        # hipGraph replay == eager launches, bit for bit
        for a, b in zip(runs[True][s], runs[False][s]):
            assert torch.equal(a, b), f"graph replay differs from eager at step {s}"
        rel = (got - ref_logits).norm(dim=-1) / ref_logits.norm(dim=-1)
        guided = OD.cfg_and_mask(cfg, ref_logits.view(2 * B, C, -1).clone(), 3.0, False, 0.8)
        ref_pred = guided.reshape(B * C, -1).argmax(-1).view(B, -1)
        gg = OD.cfg_and_mask(cfg, got.view(2 * B, C, -1).clone(), 3.0, False, 0.8)
        assert torch.equal(pred, gg.reshape(B * C, -1).argmax(-1).view(B, -1)), s      # the sampler itself is exact
        mask_rows = (masks == ref_mask).all(-1).float()                 # [layers, rows]
        topk_rows = (topk == ref_topk).float()
        first_bad = [int((mask_rows[:, r] == 0).nonzero()[0]) if bool((mask_rows[:, r] == 0).any()) else Lyr for r in range(2 * B)]
        extra = {}
        if ref32 is not None:
            c_logits, c_mask = ref32[s]
            cn = c_logits.norm(dim=-1)
            extra = {"vs_fp32_hip_logit_rel_median": float(((got - c_logits).norm(dim=-1) / cn).median()),
                     "vs_fp32_oracle_logit_rel_median": float(((ref_logits - c_logits).norm(dim=-1) / cn).median()),
                     "vs_fp32_hip_mask_agree": float((masks == c_mask).all(-1).float().mean()),
                     "vs_fp32_oracle_mask_agree": float((ref_mask == c_mask).all(-1).float().mean())}
        rec["per_step"].append({**extra, "logit_rel_median": float(rel.median()), "logit_rel_max": float(rel.max()),
                                "argmax_agree": float((pred == ref_pred).float().mean()),
                                "router_mask_agree": float(mask_rows.mean()), "router_topk_agree": float(topk_rows.mean()),
                                "router_mask_agree_by_depth": [round(float(mask_rows[i:i + 6].mean()), 4) for i in range(0, Lyr, 6)],
                                "rows_identical_routing_all_layers": int(sum(1 for f in first_bad if f == Lyr))})
    print("\nFULL-DEPTH PARITY", json.dumps(rec))
    _save(f"batch{B}" + ("_ragged" if ragged else ""), rec)
    med = max(p["logit_rel_median"] for p in rec["per_step"])
    mx = max(p["logit_rel_max"] for p in rec["per_step"])
    agree = min(p["argmax_agree"] for p in rec["per_step"])
    mask_agree = min(p["router_mask_agree"] for p in rec["per_step"])
    # bounds = first measurement on MI355X (gpurun_out/full_depth_parity.json, copied to profiles/) + margin; see DESIGN.md 2
    assert med < BOUNDS["med"] and mx < BOUNDS["max"], (med, mx)
    assert agree > BOUNDS["argmax"] and mask_agree > BOUNDS["mask"], (agree, mask_agree)
    if ref32 is not None:
        # no further from the fp32 centre than the CPU oracle is (x 1.25), in logits and in routing decisions
        for p in rec["per_step"]:
            assert p["vs_fp32_hip_logit_rel_median"] < 1.25 * p["vs_fp32_oracle_logit_rel_median"] + 0.005, p
            assert p["vs_fp32_hip_mask_agree"] > p["vs_fp32_oracle_mask_agree"] - 0.05, p


# first measurement on MI355X (36 layers, 16 rows, 64-token prompt, 3 steps): logits median 0.053-0.065 / max 0.10, arg-max codes
# 0.885-0.896, router masks 0.863-0.894 of (layer, row) pairs identical (0.93 in the first six layers, 0.83-0.86 from layer 12 on:
# every flipped near-tie sends a row through another expert and the difference travels on); bounds = that + margin
BOUNDS = {"med": 0.09, "max": 0.20, "argmax": 0.82, "mask": 0.80}


# ------------------------------------------------------------------------------------------------------------------------------------
# Every layer in isolation (reference layer: utils/UniMoE_Audio_model.py:210-256).  The test above lets differences compound over 36
# layers (every flipped near-tie travels on); here layer l of the HIP path gets the ORACLE's input of layer l -- residual stream and KV
# cache -- through the engine's per-layer probe (umoe_engine_set_probe), so each layer's weights and kernels are checked at the per-op
# tolerance on their own, router integers exactly given the GPU's own logits, and the distance from an fp32 walk along the same
# trajectory is split into the attention half (x1 - x_in) and the MoE half (x_out - x1) of every layer.
PER_LAYER_RESULTS = os.path.join(ROOT, "gpurun_out", "per_layer_parity.json")
# bounds = first measurement on MI355X + margin (profiles/r03_per_layer_parity.json: attention half 0.0003 mean / 0.0006 max, MoE half 0.004 /
# 0.019 -- the one layer at 0.019 has a row whose bf16 router logit differs in its last bit, which moves its routing WEIGHTS by ~1 % --,
# routing masks identical on every one of the 36 x 16 (layer, row) pairs, 0.03 % of the elements outside the per-op tolerance (1.1 % in
# that one layer); against the fp32 walk the HIP path and the CPU oracle sit at the same distance to four digits in both halves)
PER_LAYER_BOUNDS = {"attn_half": 0.003, "moe_half": 0.04, "mask_agree_mean": 0.97, "mask_agree_min": 0.80, "elem_viol": 0.03}


@pytest.mark.parametrize("B,ragged", [(8, False), (1, True)])
def test_full_depth_every_layer_in_isolation_vs_oracle(full_model, B, ragged, monkeypatch):
    monkeypatch.setenv("UMOE_DENSE_MIN_ROWS", "6" if ragged else "2")
    from oracle import decode as OD
    from oracle import router as OR
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.model import DecodeEngine
    cfg, gm, w, dev = full_model
    T, MAXT = 64, 48
    C, E, Lyr, D = cfg.codec_channels, cfg.num_experts, cfg.num_hidden_layers, cfg.hidden_size
    KVH, hd = cfg.num_key_value_heads, cfg.head_dim
    rows = 2 * B
    g = torch.Generator().manual_seed(200 + B)
    ids = torch.randint(0, 290, (rows, T), generator=g)
    am = torch.ones(rows, T, dtype=torch.long)
    for r in range(0, rows, 2):
        am[r, : 9 + (r % 5)] = 0
    n_codec = 24
    ids[:, -n_codec - 3:-3] = cfg.codec_placeholder_value
    codec = torch.randint(0, 1024, (rows * n_codec, C), generator=g)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    step0 = min(psteps) - 1
    forced = torch.randint(0, 1024, (B, max(pre.shape[1], step0 + 3), C), generator=g).to(torch.int32)
    keep = pre.to(torch.int32) != -1
    forced[:, : pre.shape[1]][keep] = pre.to(torch.int32)[keep]
    # ---- oracle: prefill, then ONE decode step with every layer's states kept
    tm = OD.TextModelOracle(cfg, w)
    key_valid = am.bool()
    pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
    x = OD.input_embedding(cfg, w, ids, codec)
    kv1 = torch.cat([key_valid, torch.ones((rows, 1), dtype=torch.bool)], -1)
    p1 = (kv1.long().cumsum(-1) - 1).masked_fill(~kv1, 1)[:, -1:]
    tok2 = forced[:, step0: step0 + 1].long().repeat_interleave(2, dim=0)
    with torch.no_grad():
        _, cache, _ = tm.forward(x, key_valid, pos, None)
        _, _, lay = tm.forward(OD.codec_embedding(cfg, w, tok2), kv1, p1, cache, collect_router=True)
    x_in = torch.stack([r["x_in"][:, 0] for r in lay])             # [layers, rows, D] bf16: the teacher's layer inputs
    x1_o = torch.stack([r["x1"][:, 0] for r in lay]).float()
    xo_o = torch.stack([r["x_out"][:, 0] for r in lay]).float()
    mask_o = torch.stack([r["expert_mask"].reshape(rows, E) for r in lay])
    # ---- fp32 centre: the same layers in fp32 on the SAME inputs (the bf16 oracle's layer inputs and KV cache, upcast)
    w32 = {k: v.float() for k, v in w.items()}
    tm32 = OD.TextModelOracle(cfg, w32)
    with torch.no_grad():
        c32 = [(k.float(), v.float()) for k, v in cache]
        _, _, lay32 = tm32.forward(OD.codec_embedding(cfg, w32, tok2), kv1, p1, c32, collect_router=True, layer_inputs=[t[:, None].float() for t in x_in])
    x1_c = torch.stack([r["x1"][:, 0] for r in lay32])
    xo_c = torch.stack([r["x_out"][:, 0] for r in lay32])
    mask_c = torch.stack([r["expert_mask"].reshape(rows, E) for r in lay32])
    del w32, tm32, c32, lay32
    # ---- engine: its own prefill (positions, KV slots), then the ORACLE's KV cache of every layer in its place, one eager probed step
    xg = gm.calculate_input_embedding(ids.to(dev), codec.to(dev))
    eng = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64)
    eng.prefill(xg.reshape(-1, D).contiguous(), am.to(dev))
    Lmax = eng.Lmax
    for name, idx in (("k_cache", 0), ("v_cache", 1)):
        full = torch.zeros(Lyr, rows, KVH, Lmax, hd, dtype=torch.bfloat16)
        full[:, :, :, :T] = torch.stack([c[idx] for c in cache])
        eng.write_buffer(name, full.to(dev))
    eng.start_decode(forced, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.0, top_p=1.0, top_k=45, eos_mul=0.8, do_sample=False)
    pr = eng.set_probe(teach_x=x_in.to(dev), dump_x1=True, dump_x=True, dump_logits=True)
    eng.step(use_graph=False)
    torch.cuda.synchronize()
    assert eng.handoff_error() == 0
    x1_h, xo_h, lg_h = pr["x1"].cpu().float(), pr["x"].cpu().float(), pr["logits"].cpu()
    mask_h = eng.copy_buffer("all_mask", torch.int32, (Lyr, rows, E)).cpu()
    topk_h = eng.copy_buffer("all_topk", torch.int64, (Lyr, rows)).cpu()
    eng.set_probe()
    eng.close()
    xin = x_in.float()
    rec = {"rows": rows, "layers": Lyr, "ragged": ragged, "per_layer": []}
    fro = lambda t: float(t.norm())
    for l in range(Lyr):
        # router integers: exact given the GPU's own logits (C oracle, oracle/router_oracle.c)
        o = OR.route(lg_h[l], cfg.num_dyn, cfg.mlp_dynamic_expert_num, cfg.mlp_fixed_expert_num, float(cfg.mlp_dynamic_top_p),
                     int(cfg.mlp_dynamic_top_k), float(cfg.router_jitter_noise), None)
        assert torch.equal(o["expert_mask"].to(torch.int32), mask_h[l]) and torch.equal(o["top_k"], topk_h[l]), f"layer {l}: router integers differ from the oracle on the GPU's logits"
        same = (mask_h[l] == mask_o[l].to(torch.int32)).all(-1)
        same_c = same & (mask_c[l].to(torch.int32) == mask_h[l]).all(-1)
        att = fro(x1_h[l] - x1_o[l]) / fro(x1_o[l] - xin[l])
        d_h, d_o = (xo_h[l] - x1_h[l]), (xo_o[l] - x1_o[l])
        moe = fro(d_h[same] - d_o[same]) / max(fro(d_o[same]), 1e-9) if bool(same.any()) else 0.0
        # element-wise, rows with identical routing: |hip - oracle| <= 2^-6 |oracle| + 2^-8 (the per-op bf16 tolerance)
        viol = float(((xo_h[l][same] - xo_o[l][same]).abs() > 2 ** -6 * xo_o[l][same].abs() + 2 ** -8).float().mean()) if bool(same.any()) else 0.0
        d_c = xo_c[l] - x1_c[l]
        rec["per_layer"].append({
            "attn_half_rel": att, "moe_half_rel_same_routing": moe, "mask_agree": float(same.float().mean()), "elem_viol_frac": viol,
            # distance from the fp32 walk, per half, HIP vs CPU oracle
            "attn_vs_fp32_hip": fro(x1_h[l] - x1_c[l]) / fro(x1_c[l] - xin[l]), "attn_vs_fp32_oracle": fro(x1_o[l] - x1_c[l]) / fro(x1_c[l] - xin[l]),
            "moe_vs_fp32_hip": fro(d_h[same_c] - d_c[same_c]) / max(fro(d_c[same_c]), 1e-9) if bool(same_c.any()) else None,
            "moe_vs_fp32_oracle": fro(d_o[same_c] - d_c[same_c]) / max(fro(d_c[same_c]), 1e-9) if bool(same_c.any()) else None})
    pl = rec["per_layer"]
    mean = lambda k: sum(p[k] for p in pl if p[k] is not None) / max(1, sum(1 for p in pl if p[k] is not None))
    rec["summary"] = {k: {"mean": mean(k), "max": max(p[k] for p in pl if p[k] is not None)} for k in pl[0]}
    rec["summary"]["extra_distance_from_fp32"] = {"attention_half": mean("attn_vs_fp32_hip") - mean("attn_vs_fp32_oracle"),
                                                  "moe_half": mean("moe_vs_fp32_hip") - mean("moe_vs_fp32_oracle")}
    print("\nPER-LAYER PARITY", json.dumps(rec["summary"]))
    os.makedirs(os.path.dirname(PER_LAYER_RESULTS), exist_ok=True)
    allr = json.load(open(PER_LAYER_RESULTS)) if os.path.exists(PER_LAYER_RESULTS) else {}
    allr[f"batch{B}" + ("_ragged" if ragged else "")] = rec
    json.dump(allr, open(PER_LAYER_RESULTS, "w"), indent=1)
    s = rec["summary"]
    Bd = PER_LAYER_BOUNDS
    assert s["attn_half_rel"]["max"] < Bd["attn_half"], s["attn_half_rel"]
    assert s["moe_half_rel_same_routing"]["max"] < Bd["moe_half"], s["moe_half_rel_same_routing"]
    assert s["elem_viol_frac"]["max"] < Bd["elem_viol"], s["elem_viol_frac"]
    assert s["mask_agree"]["mean"] > Bd["mask_agree_mean"] and min(p["mask_agree"] for p in pl) >= (Bd["mask_agree_min"] if rows >= 8 else 0.0), s["mask_agree"]
    # no half of any layer further from the fp32 walk than the CPU oracle's, x 1.25 (averaged over the layers)
    assert mean("attn_vs_fp32_hip") < 1.25 * mean("attn_vs_fp32_oracle") + 0.002, s
    assert mean("moe_vs_fp32_hip") < 1.25 * mean("moe_vs_fp32_oracle") + 0.002, s
