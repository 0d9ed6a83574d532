"""Headline benchmark: audio-tokens/s of the TTS decode loop (BASELINE.json configs[1]).

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" = one decode step of the whole batch: 36 x (RMSNorm, QKV, mRoPE, KV-cached GQA attention, o_proj, RMSNorm,
Top-P router, ragged dispatch, grouped SwiGLU experts + shared experts, combine) + codec head + CFG + top-k/top-p
sampling + EOS/delay bookkeeping, for batch 8 = 16 CFG rows, on synthetic N(0, 0.02^2) bf16 weights of the full
utils/config.json architecture.  One audio token = one generated frame of one sequence (12 codebook ids).
Inputs (weights, prompt KV cache) are resident in HBM when the timed region starts; prefill is reported separately.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: grouped gate/up SwiGLU GEMM, HBM-bound) and
`cpu_baseline` (the CPU oracle timed on the host cores, a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=8, help="sequences per GPU (CFG doubles the rows)")
    ap.add_argument("--prompt", type=int, default=300)
    ap.add_argument("--layers", type=int, default=0, help="debug only: override num_hidden_layers (0 = 36)")
    ap.add_argument("--codec-channels", type=int, default=0,
                    help="0 = the reference's 12 channels (16 kHz DAC, 50 frames/s); 9 = the 44.1 kHz reading of BASELINE (SURVEY 8d: delay [0,8..15], 86.1 frames/s)")
    ap.add_argument("--parallel", default="auto", choices=["auto", "ep", "replica"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5)
    return ap.parse_args()


def synth_prompt(cfg, B, T, device):
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(0, 151643, (2 * B, T), generator=g)
    am = torch.ones(2 * B, T, dtype=torch.long)
    n_codec = min(245, T - 8)
    ids[:, -n_codec - 3:-3] = cfg.codec_placeholder_value
    am[0::2, :17] = 0                                  # negative prompts are shorter: left padding on uncond rows
    g.manual_seed(7)
    codec = torch.randint(0, 1024, (2 * B * n_codec, cfg.codec_channels), generator=g)
    return ids.to(device), am.to(device), codec.to(device)


def gpu_run(args, rank, world, device):
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.config import UniMoEAudioConfig
    from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration

    cfg = UniMoEAudioConfig()
    if args.layers:
        cfg.num_hidden_layers = args.layers
    if args.codec_channels:
        cfg.codec_channels = args.codec_channels
        cfg.codec_delay_pattern = [0] + list(range(8, 8 + args.codec_channels - 1))
    B, T, K, W = args.batch, args.prompt, args.steps, args.warmup
    t0 = time.time()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(device):
        model = UniAudioRVQQwen2_5VLMoEForConditionalGeneration(cfg)
    torch.set_default_dtype(torch.float32)
    model.init_synthetic(1234).eval()
    torch.cuda.synchronize()
    t_build = time.time() - t0
    max_tokens = K + W + 64
    eng = model.engine(B, T, max_tokens, attn_splits=int(os.environ.get("UMOE_ATTN_SPLITS", "8")))   # (experiment knob; 8 measured best)
    ids, am, codec = synth_prompt(cfg, B, T, device)
    x = model.calculate_input_embedding(ids, codec)
    torch.cuda.synchronize()
    t1 = time.time()
    eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am)
    torch.cuda.synchronize()
    t_prefill = time.time() - t1
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    # reference generate() defaults (model.py:1083-1088); EOS disabled (min_tokens > steps) so step counts are fixed
    eng.start_decode(pre, psteps, max_tokens, max_tokens, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8,
                     do_sample=True, seed=1234 + rank)
    use_graph = not args.no_graph
    for _ in range(W):
        eng.step(use_graph)
    barrier(world)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(K):
        eng.step(use_graph)
    torch.cuda.synchronize()
    barrier(world)
    dt = time.perf_counter() - t2
    # which experts were hit (last step, every layer) -> algorithmic bytes of the dominant kernel
    E = cfg.num_experts
    masks = eng.copy_buffer("all_mask", torch.int32, (cfg.num_hidden_layers, 2 * B, E)).cpu()
    hit = (masks[:, :, : cfg.mlp_dynamic_expert_num].sum(1) > 0).sum(1).float()          # |U_l| per layer
    topk = eng.copy_buffer("all_topk", torch.int64, (cfg.num_hidden_layers, 2 * B)).cpu().float()
    prof = eng.profile_steps(4)
    info = dict(t_build=t_build, t_prefill=t_prefill, mean_experts_hit=float(hit.mean()), mean_top_k=float(topk.mean()),
                prof=prof, cfg=cfg, kv_len_end=T + W + K + 4)
    return dt, info


def barrier(world):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def roofline(info):
    cfg = info["cfg"]
    D, Id, Is = cfg.hidden_size, cfg.dynamic_intermediate_size, cfg.shared_intermediate_size
    U = info["mean_experts_hit"]
    # algorithmic bytes of ONE launch of the grouped gate/up SwiGLU kernel: gate+up weights of every
    # routed expert hit + of the shared experts, bf16; activations (16 x 2048) and outputs are < 0.3 % and left out
    bytes_per_launch = (U * 2 * Id * D + cfg.mlp_fixed_expert_num * 2 * Is * D) * 2.0
    ms, n = info["prof"]["gateup"]
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    kname = "wstream_gemm<14, 1, 0, 2, 8, true>"
    traffic = None
    try:  # HBM bytes per launch from the committed rocprofv3 --pmc passes of this same command (profiles/)
        with open(os.path.join(ROOT, "profiles", "r01h_pmc_traffic.json")) as f:
            traffic = json.load(f)["kernels"][kname]["traffic_bytes"]
    except Exception:
        pass
    return {"bound": "hbm", "kernel": kname + " (grouped gate/up SwiGLU, 8 routed + 2 shared experts in one launch; 16 of its tile-less workgroups run the Top-P router)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(ms * 1e3, 2),
            "launches_per_step": n, "experts_hit_per_layer": round(U, 2),
            "note": "achieved = algorithmic bytes / HIP-event interval on the launch stream (eager profiling pass right after the "
                    "timed region; the interval includes the launch gap, rocprofv3 durations are in profiles/r01h_decode_kernels.md); "
                    "traffic = FETCH_SIZE*2 + WRITE_SIZE per launch from separate rocprofv3 --pmc passes (profiles/r01h_pmc_traffic.md)"}


def cpu_baseline(args):
    """The CPU oracle (oracle/decode.py, reference-like torch-CPU path) on a bounded sample: `cpu_steps` decode steps of
    the full 36-layer model at batch 8 (16 rows) with a synthetic 300-token KV cache, greedy sampling."""
    from oracle import decode as OD
    from unimoe_audio_amd.config import UniMoEAudioConfig
    cfg = UniMoEAudioConfig()
    if args.layers:
        cfg.num_hidden_layers = args.layers
    if args.codec_channels:
        cfg.codec_channels = args.codec_channels
        cfg.codec_delay_pattern = [0] + list(range(8, 8 + args.codec_channels - 1))
    torch.manual_seed(0)
    bf = torch.bfloat16
    D, H, KV, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    w = {}

    proto = {}

    def rnd(*shape):
        # one N(0, 0.02^2) tensor per distinct shape, cloned for every further use: every weight keeps its own 14 GB of
        # storage (realistic cache behaviour) while generation stays cheap; router gates below are drawn independently
        if shape not in proto:
            proto[shape] = (torch.randn(*shape) * 0.02).to(bf)
            return proto[shape]
        return proto[shape].clone()
    for l in range(cfg.num_hidden_layers):
        p = f"language_model.layers.{l}."
        w[p + "input_layernorm.weight"] = torch.ones(D, dtype=bf)
        w[p + "post_attention_layernorm.weight"] = torch.ones(D, dtype=bf)
        for n, o in (("q", H * hd), ("k", KV * hd), ("v", KV * hd)):
            w[p + f"self_attn.{n}_proj.weight"] = rnd(o, D)
            w[p + f"self_attn.{n}_proj.bias"] = torch.zeros(o, dtype=bf)
        w[p + "self_attn.o_proj.weight"] = rnd(D, H * hd)
        w[p + "mlp.gate.weight"] = (torch.randn(cfg.num_experts, D) * 0.02).to(bf)
        for e in range(cfg.mlp_dynamic_expert_num):
            q = p + f"mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.{e}."
            w[q + "gate_proj.weight"], w[q + "up_proj.weight"] = rnd(cfg.dynamic_intermediate_size, D), rnd(cfg.dynamic_intermediate_size, D)
            w[q + "down_proj.weight"] = rnd(D, cfg.dynamic_intermediate_size)
        for i in range(cfg.mlp_fixed_expert_num):
            q = p + f"mlp.fixed_real_moe.{i}."
            w[q + "gate_proj.weight"], w[q + "up_proj.weight"] = rnd(cfg.shared_intermediate_size, D), rnd(cfg.shared_intermediate_size, D)
            w[q + "down_proj.weight"] = rnd(D, cfg.shared_intermediate_size)
    w["language_model.norm.weight"] = torch.ones(D, dtype=bf)
    for c in range(cfg.codec_channels):
        w[f"codec_embed_tokens.{c}.weight"] = rnd(cfg.codec_vocab_size, D)
    w["codec_head.weight"] = rnd(cfg.codec_channels * cfg.codec_vocab_size, D)
    B, L = args.batch, args.prompt
    rows = 2 * B
    tm = OD.TextModelOracle(cfg, w)
    cache = [(torch.randn(rows, KV, L, hd).to(bf), torch.randn(rows, KV, L, hd).to(bf)) for _ in range(cfg.num_hidden_layers)]
    key_valid = torch.ones(rows, L, dtype=torch.bool)
    tok = torch.randint(0, 1024, (B, 1, cfg.codec_channels))
    times = []
    with torch.no_grad():
        for s in range(args.cpu_steps + 1):
            t0 = time.perf_counter()
            key_valid = torch.cat([key_valid, torch.ones(rows, 1, dtype=torch.bool)], -1)
            pos = (key_valid.long().cumsum(-1) - 1)[:, -1:]
            h, cache, _ = tm.forward(OD.codec_embedding(cfg, w, tok.repeat_interleave(2, dim=0)), key_valid, pos, cache)
            logits = torch.nn.functional.linear(h, w["codec_head.weight"]).float().view(rows, -1, cfg.codec_channels, cfg.codec_vocab_size)[:, -1]
            guided = OD.cfg_and_mask(cfg, logits, 3.0, False, 0.8)
            pred = OD.sample_next_token(guided.reshape(B * cfg.codec_channels, -1), 1.2, 0.95, 45, cfg.codec_eos_value)
            tok = pred.view(B, 1, cfg.codec_channels)
            times.append(time.perf_counter() - t0)
    steady = sorted(times[1:])                           # first step = warm-up
    sec = steady[len(steady) // 2]                       # median
    return {"value": round(B / sec, 3), "unit": "audio-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"median of {len(steady)} decode steps (after 1 warm-up) of the full {cfg.num_hidden_layers}-layer model, batch {B} "
                      f"(16 CFG rows), synthetic KV cache of {L} tokens, torch-CPU bf16 oracle (oracle/decode.py), "
                      f"{sec * 1e3:.0f} ms/step"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    mode = args.parallel
    dist_on = world > 1 or "RANK" in os.environ
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        try:
            dist.init_process_group("nccl", device_id=device)          # "nccl" is RCCL on ROCm
            probe = torch.ones(1, device=device)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
        except Exception as e:  # keep the scaling run alive: the replica mode only needs a barrier and a MAX of one scalar
            print(f"[bench] RCCL init failed ({e!r}); falling back to gloo for the barrier/reduce", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
            dist.init_process_group("gloo")
    if world > 1:
        if mode in ("auto", "ep"):
            # the in-engine RCCL exchange is the next step: this round every rank runs a full replica
            mode = "replica"
    else:
        mode = "single"
    dt, info = gpu_run(args, rank, world, device)
    if dist_on:
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        t = torch.tensor([dt], device=device if on_gpu else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                        # the slowest rank defines the step time
        dt = float(t.item())
    B, K, W = args.batch, args.steps, args.warmup
    if rank == 0:
        value = world * B * K / dt
        cfg = info["cfg"]
        out = {
            "metric": "audio-tokens/sec/node (TTS decode, bs=8 per GPU)", "value": round(value, 2), "unit": "audio-tokens/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: TTS decode batch 8 (16 CFG rows) per GPU, 300-token prompt, "
                                   f"{cfg.num_hidden_layers}-layer DCMoE (8+1 routed, 2 shared), 12x1027 codec head, "
                                   "CFG 3.0 + top-k 45 + top-p 0.95 sampling, 16 kHz DAC 50 frames/s",
                       "global_batch": world * B, "prompt_len": args.prompt, "kv_len_end": info["kv_len_end"],
                       "parallelism": mode + (str(world) if world > 1 else ""), "graph": not args.no_graph,
                       "codes_per_s": round(value * cfg.codec_channels, 1), "prefill_s": round(info["t_prefill"], 3),
                       "mean_top_k": round(info["mean_top_k"], 2)},
            "roofline": roofline(info),
            "kernel_ms_per_step": {k: round(v[0] * v[1], 4) for k, v in info["prof"].items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
                out["config"]["gpu_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            except Exception as e:  # the GPU number must not be lost to a host-side problem
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist_on:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
