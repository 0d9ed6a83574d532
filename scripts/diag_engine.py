"""Diagnostic (not a test): per-step comparison of the GPU engine with the CPU oracle under teacher forcing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import decode as OD
import test_gpu_engine as TE
dev = torch.device("cuda:0")
cfg = TE.small_cfg()
m, w = TE.build(cfg, 1, 0.06)
B, T, steps = 2, 12, 12
ids, am, codec = TE.prompt(cfg, B, T, 2, [3, 0, 1, 0])
pre, psteps = OD.prepare_audio_prompt(cfg, [None] * B)
gen = OD.GenerateOracle(cfg, w)
gen.generate(ids, am, pre, psteps, 60, 6, codec_input_ids=codec, cfg_scale=3.0, do_sample=False, eos_prob_mul_factor=0.8)
otok = gen.tokens
tm = OD.TextModelOracle(cfg, w)
key_valid = am.bool()
pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
x = OD.input_embedding(cfg, w, ids, codec)
hpre, cache, rpre = tm.forward(x, key_valid, pos, None, collect_router=True)
gm = m.to(dev)
eng = gm.engine(B, T, 60)
xg = gm.calculate_input_embedding(ids.to(dev), codec.to(dev))
eng.prefill(xg.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev))
# compare KV cache after prefill, layer by layer
kc = eng.copy_buffer("k_cache", torch.bfloat16, (cfg.num_hidden_layers, 2*B, cfg.num_key_value_heads, eng.Lmax, 128)).cpu()
for l in range(cfg.num_hidden_layers):
    kref = cache[l][0]
    sel = am.bool().unsqueeze(1).expand(-1, cfg.num_key_value_heads, -1)
    d = kc[l][:, :, :T].float()[sel] - kref.float()[sel]
    print(f"prefill layer {l} K rel err {float(d.norm()/kref.float()[sel].norm()):.5f} max {float(d.abs().max()):.4f}")
forced = otok.to(torch.int32).clone()
eng.start_decode(forced, psteps, 60, 6, cfg_scale=3.0, temperature=1.0, top_p=1.0, top_k=45, eos_mul=0.8, do_sample=False)
E = cfg.num_experts
for s in range(steps):
    kv1 = torch.cat([key_valid, torch.ones((2 * B, 1), dtype=torch.bool)], -1)
    p1 = (kv1.long().cumsum(-1) - 1).masked_fill(~kv1, 1)[:, -1:]
    tok2 = otok[:, s: s + 1].repeat_interleave(2, dim=0)
    h, cache, router = tm.forward(OD.codec_embedding(cfg, w, tok2), kv1, p1, cache, collect_router=True)
    key_valid = kv1
    ref_logits = torch.nn.functional.linear(h, w["codec_head.weight"]).float()[:, -1]
    eng.step(use_graph=False)
    got = eng.copy_buffer("logits", torch.float32, (2 * B, cfg.codec_channels * cfg.codec_vocab_size)).cpu()
    rel = (got - ref_logits).norm(dim=-1) / ref_logits.norm(dim=-1)
    guided = OD.cfg_and_mask(cfg, ref_logits.view(2 * B, cfg.codec_channels, -1).clone(), 3.0, s >= 6, 0.8)
    gg = OD.cfg_and_mask(cfg, got.view(2 * B, cfg.codec_channels, -1).clone(), 3.0, s >= 6, 0.8)
    ref_pred = guided.reshape(B * cfg.codec_channels, -1).argmax(-1).view(B, -1)
    got_pred_from_logits = gg.reshape(B * cfg.codec_channels, -1).argmax(-1).view(B, -1)
    pred = eng.copy_buffer("pred", torch.int64, (B, cfg.codec_channels)).cpu()
    masks = eng.copy_buffer("all_mask", torch.int32, (cfg.num_hidden_layers, 2 * B, E)).cpu()
    topk = eng.copy_buffer("all_topk", torch.int64, (cfg.num_hidden_layers, 2 * B)).cpu()
    magree = [(masks[l] == router[l]["expert_mask"]).all(-1).tolist() for l in range(cfg.num_hidden_layers)]
    top2 = guided.reshape(B*cfg.codec_channels, -1).topk(2, -1).values
    gap = (top2[:,0]-top2[:,1]).view(B,-1)
    print(f"step {s}: rel {[round(float(r),4) for r in rel]} mask_agree {magree} kernel_vs_cpuargmax_of_gpu_logits {int((pred==got_pred_from_logits).sum())}/{pred.numel()} vs_oracle {int((pred==ref_pred).sum())}/{pred.numel()} min_gap {float(gap.min()):.3f}")
    if s == 0:
        print(" pred", pred.tolist()); print(" ref ", ref_pred.tolist()); print(" gpuL", got_pred_from_logits.tolist())
