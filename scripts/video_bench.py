"""BASELINE configs[4] measurement (not the headline bench): video_text_to_music, batch 2 (4 CFG rows), one 8-frame clip per row at the
reference's pixel budget (64 * 28 * 28 px per frame, mod.py:49-53 -> 4 x 16 x 16 patches = 256 video tokens per row), full multimodal
path on ONE MI355X: vision tower (32 blocks, 1280-d, window / full attention, patch merger) -> video embeddings scattered over the
<|video_pad|> tokens -> 3-D mRoPE index -> engine prefill with explicit positions -> K decode steps through the captured step graph.
Synthetic N(0, 0.02^2) weights of the utils/config.json architecture (text model AND vision tower), random pixels.  The reference
quotes this config on 8 GPUs; two sequences shard over at most two (a cond / uncond pair stays on one GPU), so the number here is the
one-GPU one.  Prints one JSON line.

  python scripts/video_bench.py [--steps 500] [--warmup 10]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unimoe_audio_amd.codec_utils import prepare_audio_prompt
from unimoe_audio_amd.config import UniMoEAudioConfig
from unimoe_audio_amd.model import DecodeEngine, UniAudioRVQQwen2_5VLMoEForConditionalGeneration

# utils/config.json:159-183
VISION = dict(depth=32, fullatt_block_indexes=[7, 15, 23, 31], hidden_act="silu", hidden_size=1280, in_chans=3, intermediate_size=3420,
              num_heads=16, out_hidden_size=2048, patch_size=14, spatial_merge_size=2, temporal_patch_size=2, tokens_per_second=2,
              window_size=112)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=500)       # 10 s of 50 Hz frames
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--frames", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = UniMoEAudioConfig()
    cfg.vision_config = dict(VISION)
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(dev):
        model = UniAudioRVQQwen2_5VLMoEForConditionalGeneration(cfg)
    torch.set_default_dtype(torch.float32)
    model.init_synthetic(1234).eval()
    g = torch.Generator(device=dev).manual_seed(4321)
    with torch.no_grad():
        for n, p in model.visual.named_parameters():
            if p.dim() == 1 and n.endswith(".weight"):            # RMSNorm weights (blocks' norm1 / norm2, merger ln_q)
                p.fill_(1.0)
            elif n.endswith(".bias"):
                p.zero_()
            else:
                p.normal_(0, 0.02, generator=g)
    B, K, W = a.batch, a.steps, a.warmup
    rows = 2 * B
    tt, gh, gw = a.frames // VISION["temporal_patch_size"], 16, 16
    n_vid = tt * gh * gw // (VISION["spatial_merge_size"] ** 2)                   # 256 video tokens per row
    n_text0, n_text1 = 20, 22
    T = n_text0 + 1 + n_vid + 1 + n_text1                                        # 300 tokens
    gen = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 151643, (rows, T), generator=gen)
    ids[:, n_text0] = cfg.vision_start_token_id
    ids[:, n_text0 + 1: n_text0 + 1 + n_vid] = cfg.video_token_id
    ids[:, n_text0 + 1 + n_vid] = cfg.vision_end_token_id
    am = torch.ones(rows, T, dtype=torch.long)
    am[0::2, :9] = 0                                                              # shorter negative prompts: left padding
    grid = torch.tensor([[tt, gh, gw]] * rows)
    patch_dim = VISION["in_chans"] * VISION["temporal_patch_size"] * VISION["patch_size"] ** 2
    px = torch.randn(rows * tt * gh * gw, patch_dim, generator=gen).to(torch.bfloat16).to(dev)
    sec = torch.tensor([1.0] * rows)
    ids_d, am_d = ids.to(dev), am.to(dev)

    def embed():
        return model.multimodal_embedding(ids_d, None, pixel_values_videos=px, video_grid_thw=grid)

    with torch.no_grad():
        xg = embed()                                                              # (first call: workspace allocation)
        torch.cuda.synchronize()
        tv = []
        for _ in range(5):
            t0 = time.perf_counter()
            xg = embed()
            torch.cuda.synchronize()
            tv.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    pos, delta = model.get_rope_index(ids, None, grid, sec, am)
    t_rope = time.perf_counter() - t0
    max_tokens = K + W + 64
    eng = DecodeEngine(model, B, Lmax=T + max_tokens + 8, Tmax=max_tokens + 64)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.prefill(xg.reshape(-1, cfg.hidden_size).contiguous(), am_d, position_ids=pos, rope_deltas=delta)
    torch.cuda.synchronize()
    t_prefill = time.perf_counter() - t0
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    eng.start_decode(pre, psteps, max_tokens, max_tokens, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8,
                     do_sample=True, seed=1234)
    for _ in range(W):
        eng.step(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        eng.step(True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    hand = eng.handoff_error()
    eng.close()
    if hand:
        raise SystemExit(f"video_bench: an in-launch hand-off of the decode engine timed out (code {hand})")
    # vision tower FLOPs (forward): per patch token per block 2 * (4 d^2 + 3 d i) + attention, + patch embedding + merger
    d, i = VISION["hidden_size"], VISION["intermediate_size"]
    n_patch = rows * tt * gh * gw
    flop = n_patch * VISION["depth"] * 2 * (4 * d * d + 3 * d * i) + n_patch * 2 * patch_dim * d + (n_patch // 4) * 2 * (4 * d * 4 * d + 4 * d * 2048)
    tvm = sorted(tv)[len(tv) // 2]
    print(json.dumps({
        "workload": f"BASELINE configs[4]: video_text_to_music, batch {B} ({rows} CFG rows), {a.frames}-frame clip per row = {n_vid} video tokens, "
                    f"{T}-token prompt, {cfg.num_hidden_layers}-layer DCMoE + 32-block vision tower, 1 x MI355X, bf16, synthetic weights",
        "vision_tower_ms": round(tvm * 1e3, 3), "vision_patches": n_patch, "vision_tflops": round(flop / tvm / 1e12, 1),
        "rope_index_host_ms": round(t_rope * 1e3, 3), "prefill_ms": round(t_prefill * 1e3, 3),
        "decode_steps": K, "decode_ms_per_step": round(dt / K * 1e3, 4), "audio_tokens_per_s": round(B * K / dt, 2),
        "time_to_first_frame_ms": round((tvm + t_rope + t_prefill + dt / K) * 1e3, 2),
        "end_to_end_s_for_10s_of_music": round(tvm + t_rope + t_prefill + 500 * dt / K, 3)}))


if __name__ == "__main__":
    main()
