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
    ap.add_argument("--parallel", default="auto", choices=["auto", "ep", "replica"],
                    help="N > 1: ep = expert-parallel decode (BASELINE configs[3] layout: batch-sharded rows, n_real/N routed experts per "
                         "GPU, exchange inside the step graph); replica = N independent full models; auto = both are timed (they give the "
                         "same tokens), `value` is the faster one and the other is reported beside it")
    ap.add_argument("--ep-backend", default=os.environ.get("UMOE_EP_BACKEND", "peer"), choices=["peer", "rccl"],
                    help="exchange of the expert-parallel step: xGMI peer stores (default) or RCCL calls captured in the graph")
    ap.add_argument("--ep-emulate", type=int, default=0,
                    help="single GPU only: time ONE rank of an N-rank expert-parallel job in loopback (no peers; a kernel-time proxy, "
                         "reported as a secondary field, never as `value`)")
    ap.add_argument("--ep-trial", default="", help=argparse.SUPPRESS)      # internal: the sandboxed rehearsal of an exchange backend (see main)
    ap.add_argument("--no-replica-check", action="store_true", help="N > 1, ep: skip the replica leg (secondary number + bit check)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-config1", action="store_true", help="skip the BASELINE configs[0] (single prompt) legs")
    ap.add_argument("--no-config3", action="store_true", help="skip the BASELINE configs[2] leg (one training step, scripts/train_bench.py as a child process)")
    ap.add_argument("--no-config5", action="store_true", help="skip the BASELINE configs[4] leg (video_text_to_music, scripts/video_bench.py as a child process)")
    ap.add_argument("--cpu-steps", type=int, default=7, help="timed CPU-oracle decode steps (median; two warm-up steps before them)")
    return ap.parse_args()


def synth_prompt(cfg, B, T, device, rank=0):
    g = torch.Generator().manual_seed(11 + 1000 * rank)
    ids = torch.randint(0, 151643, (2 * B, T), generator=g)
    am = torch.ones(2 * B, T, dtype=torch.long)
    n_codec = min(245, T - 8)
    ids[:, -n_codec - 3:-3] = cfg.codec_placeholder_value
    am[0::2, :17] = 0                                  # negative prompts are shorter: left padding on uncond rows
    g.manual_seed(7 + 1000 * rank)
    codec = torch.randint(0, 1024, (2 * B * n_codec, cfg.codec_channels), generator=g)
    return ids.to(device), am.to(device), codec.to(device)


def make_cfg(args):
    from unimoe_audio_amd.config import UniMoEAudioConfig
    cfg = UniMoEAudioConfig()
    if args.layers:
        cfg.num_hidden_layers = args.layers
    if args.codec_channels:
        cfg.codec_channels = args.codec_channels
        cfg.codec_delay_pattern = [0] + list(range(8, 8 + args.codec_channels - 1))
    return cfg


def build_model(cfg, device):
    from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration
    t0 = time.time()
    torch.set_default_dtype(torch.bfloat16)
    with torch.device(device):
        model = UniAudioRVQQwen2_5VLMoEForConditionalGeneration(cfg)
    torch.set_default_dtype(torch.float32)
    model.init_synthetic(1234).eval()          # same seed on every rank: expert-parallel ranks hold slices of ONE model
    torch.cuda.synchronize()
    return model, time.time() - t0


def decode_leg(model, cfg, args, device, rank, B, ep=None, profile=True, steps=None, warmup=None):
    """prefill + W warm-up steps + K timed steps of one engine; the timed region is bracketed by barrier + synchronize."""
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.model import DecodeEngine
    T = args.prompt
    K = args.steps if steps is None else steps
    W = args.warmup if warmup is None else warmup
    max_tokens = K + W + 64
    eng = DecodeEngine(model, B, Lmax=T + max_tokens + 8, Tmax=max_tokens + 64,
                       attn_splits=int(os.environ.get("UMOE_ATTN_SPLITS", "8")), ep=ep)     # (experiment knob; 8 measured best)
    ids, am, codec = synth_prompt(cfg, B, T, device, rank)
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
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(K):
        eng.step(use_graph)
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t2
    E = cfg.num_experts
    masks = eng.copy_buffer("all_mask", torch.int32, (cfg.num_hidden_layers, 2 * B, E)).cpu()
    hit = (masks[:, :, : cfg.mlp_dynamic_expert_num].sum(1) > 0).sum(1).float()          # |U_l| per layer (last step)
    topk = eng.copy_buffer("all_topk", torch.int64, (cfg.num_hidden_layers, 2 * B)).cpu().float()
    tokens = eng.tokens[:, : K + W + 2].cpu().clone()
    prof = eng.profile_steps(4) if profile else None
    expert_launch = eng.info("expert_launch")
    err = eng.ep_error() if ep is not None else 0
    hand = eng.handoff_error()
    if hand and not (ep is not None and hand == 1):
        raise SystemExit(f"bench.py: an in-launch hand-off of the decode engine timed out (code {hand}): the numbers would be of a broken run")
    if ep is not None:
        barrier()                     # peers may still be reading this rank's exchange region
    info = dict(dt=dt, t_prefill=t_prefill, mean_experts_hit=float(hit.mean()), mean_top_k=float(topk.mean()), prof=prof,
                kv_len_first=T + W, kv_len_end=T + W + K, tokens=tokens, ep_error=err, steps=K, expert_launch=expert_launch)
    eng.close()
    return info


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(v: float) -> float:
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return v
    t = torch.tensor([v], dtype=torch.float64)          # CPU tensor: the control plane runs on gloo
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def step_bytes(cfg, rows, kv_mean, experts_hit, ep=1):
    """Algorithmic HBM bytes of ONE decode step of ONE GPU: per layer attention + shared experts + gate + norms
    (52.75 MB at the reference sizes) + 33.82 MB per routed expert streamed (|U_l| of them; expert parallel: the n_real/ep local
    ones), the KV read 1 KiB per cached token per row per layer, the codec head; activations left out (< 0.1 %)."""
    D, H, KV, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    Id, Is, E = cfg.dynamic_intermediate_size, cfg.shared_intermediate_size, cfg.num_experts
    attn = (H * hd + 2 * KV * hd) * D + (H * hd + 2 * KV * hd) + D * H * hd
    per_layer = 2.0 * (attn + cfg.mlp_fixed_expert_num * 3 * Is * D + E * D + 2 * D)
    routed = 2.0 * 3 * Id * D * (experts_hit if ep == 1 else cfg.mlp_dynamic_expert_num / ep)
    kv = 2.0 * 2 * KV * hd * kv_mean * rows
    head = 2.0 * cfg.codec_channels * cfg.codec_vocab_size * D
    return cfg.num_hidden_layers * (per_layer + routed + kv) + head


PROFILE_REF = os.path.join(ROOT, "profiles", "bench_roofline_ref.json")   # rocprofv3 figures of this same command (committed)


def roofline(cfg, info, args, ep=1):
    """Dominant kernel = the grouped gate/up SwiGLU launch (HBM-bound weight streaming).  achieved = algorithmic bytes per launch
    / its average HIP-event interval on the launch stream, measured live in this run (eager pass right after the timed region)."""
    D, Id, Is = cfg.hidden_size, cfg.dynamic_intermediate_size, cfg.shared_intermediate_size
    U = info["mean_experts_hit"]
    n_fix, n_real = cfg.mlp_fixed_expert_num, cfg.mlp_dynamic_expert_num
    ms, n = info["prof"]["gateup"]
    fused = ep == 1 and info["prof"].get("down", (0.0, 0))[1] == 0      # no down-projection launch: the fused expert launch ran
    if fused:
        # gate + up + down weights of every routed expert hit and of the shared experts, bf16: both expert GEMMs of the
        # layer are ONE launch (every workgroup: gate/up slice, publish, down slice)
        bytes_per_launch = (U * 3 * Id * D + n_fix * 3 * Is * D) * 2.0
        if info.get("expert_launch") == 2:
            kname = "moe_flat_kernel"
            what = (" (gate/up SwiGLU + down projections of 8 routed + 2 shared experts in one launch of one workgroup per CU with a "
                    "byte-balanced static schedule; the first 16 workgroups also run the Top-P router of one row each)")
        else:
            kname = "moe_fused_kernel"
            what = (" (grouped gate/up SwiGLU + down projections of 8 routed + 2 shared experts in one launch; 16 of its tile-less "
                    "workgroups run the Top-P router and hand the normalised rows over)")
    elif ep == 1:
        # gate+up weights of every routed expert hit + of the shared experts, bf16; activations / outputs < 0.3 %
        bytes_per_launch = (U * 2 * Id * D + n_fix * 2 * Is * D) * 2.0
        kname = "wstream_gemm<14, 1, 0, 2, 8, true>"
        what = " (grouped gate/up SwiGLU, 8 routed + 2 shared experts in one launch; 16 of its tile-less workgroups run the Top-P router)"
    elif info.get("expert_launch") == 3:
        # expert parallel, the MoE half of a layer as ONE launch (umoe_moe_ep.hip): gate + up + down weights of the n_real / ep local experts and
        # of the shared experts, each streamed once; the exchange (riders, return stores) is inside the launch
        bytes_per_launch = ((n_real / ep) * 3 * Id * D + n_fix * 3 * Is * D) * 2.0
        kname = f"moe_ep_kernel<{ep}>"
        what = (f" (expert parallel x{ep}: local + shared experts' gate/up SwiGLU and down projections in one launch of one workgroup per CU; "
                "tile riders push / re-lay the rows, the down epilogue stores into the owners' return slabs)")
    else:
        # expert parallel, launch-per-kernel exchange: two gate/up launches per layer (shared experts beside the exchange, then the local
        # experts over the ep*16 gathered rows); algorithmic bytes = each weight once
        bytes_per_launch = ((n_real / ep) * 2 * Id * D + n_fix * 2 * Is * D) * 2.0 / 2
        # (peer / loopback exchange: the local experts run in the multi-tile launch wstream_mt over all ranks' rows; the RCCL
        #  fallback keeps the single-tile grouped launch)
        local = "wstream_gemm<14, 1, 0, 2, 8, true>" if info.get("backend") == "rccl" else "wstream_mt (local experts over every rank's rows)"
        kname = local + " + wstream_gemm<2, 1, 0, 2, 8, false>"
        what = f" (expert parallel x{ep}: local experts' and shared experts' gate/up launches, averaged)"
    achieved = bytes_per_launch / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    K = info["steps"]
    kv_mean = (info["kv_len_first"] + info["kv_len_end"] + 1) / 2.0
    sb = step_bytes(cfg, 2 * args.batch, kv_mean, U, ep)
    step_gbs = sb / (info["dt"] / K) / 1e9
    out = {"bound": "hbm", "kernel": kname + what, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None, "traffic_source": None,
           "bytes_per_launch": int(bytes_per_launch), "avg_launch_us": round(ms * 1e3, 2), "launches_per_step": n,
           "experts_hit_per_layer": round(U, 2),
           "step_bytes": int(sb), "step_achieved": round(step_gbs, 1), "step_frac": round(step_gbs / HBM_PEAK_GBS, 4),
           "note": "achieved = algorithmic bytes / HIP-event interval on the launch stream (live, eager pass after the timed region; "
                   "the interval includes the launch gap); step_* = SURVEY 8d bytes of a whole decode step (measured |U_l|, mean KV "
                   "length of the timed steps) / ms_per_step; traffic / rocprof_kernel_us are NOT measured by this run: they are "
                   "read from the committed rocprofv3 passes named in traffic_source"}
    if ep == 1:
        try:
            with open(PROFILE_REF) as f:
                ref = json.load(f)
            if ref.get("kernel", "wstream_gemm") .startswith(kname.split("<")[0]):     # (the reference must be of the kernel that ran)
                out["traffic"] = ref["traffic_bytes"]
                out["traffic_source"] = ref["source"]
                out["rocprof_kernel_us"] = ref["kernel_us"]
        except Exception:
            pass
    return out


_CPU_W = {}


def cpu_baseline(args, batch=None, steps=None):
    """The CPU oracle (oracle/decode.py, reference-like torch-CPU path) on a bounded sample: `cpu_steps` decode steps of
    the full 36-layer model at batch 8 (16 rows) with a synthetic 300-token KV cache, greedy sampling."""
    from oracle import decode as OD
    cfg = make_cfg(args)
    cpu_model, n_phys, usable = host_cpu()
    torch.set_num_threads(max(1, min(n_phys, usable)))      # one thread per PHYSICAL core: SMT siblings oversubscribe MKL
    torch.manual_seed(0)
    bf = torch.bfloat16
    D, H, KV, hd = cfg.hidden_size, cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    w = _CPU_W          # (the second call -- BASELINE configs[0], batch 1 -- reuses the 14 GB of synthetic weights)

    proto = {}

    def rnd(*shape):
        # one N(0, 0.02^2) tensor per distinct shape, cloned for every further use: every weight keeps its own 14 GB of
        # storage (realistic cache behaviour) while generation stays cheap; router gates below are drawn independently
        if shape not in proto:
            proto[shape] = (torch.randn(*shape) * 0.02).to(bf)
            return proto[shape]
        return proto[shape].clone()
    for l in range(cfg.num_hidden_layers if not w else 0):
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
    if "codec_head.weight" not in w:
        w["language_model.norm.weight"] = torch.ones(D, dtype=bf)
        for c in range(cfg.codec_channels):
            w[f"codec_embed_tokens.{c}.weight"] = rnd(cfg.codec_vocab_size, D)
        w["codec_head.weight"] = rnd(cfg.codec_channels * cfg.codec_vocab_size, D)
    B, L = (args.batch if batch is None else batch), args.prompt
    n_steps = args.cpu_steps if steps is None else steps
    rows = 2 * B
    tm = OD.TextModelOracle(cfg, w)
    cache = [(torch.randn(rows, KV, L, hd).to(bf), torch.randn(rows, KV, L, hd).to(bf)) for _ in range(cfg.num_hidden_layers)]
    key_valid = torch.ones(rows, L, dtype=torch.bool)
    tok = torch.randint(0, 1024, (B, 1, cfg.codec_channels))
    state = {"kv": key_valid, "cache": cache, "tok": tok}

    def one_step():
        t0 = time.perf_counter()
        kv = torch.cat([state["kv"], torch.ones(rows, 1, dtype=torch.bool)], -1)
        pos = (kv.long().cumsum(-1) - 1)[:, -1:]
        h, c2, _ = tm.forward(OD.codec_embedding(cfg, w, state["tok"].repeat_interleave(2, dim=0)), kv, pos, state["cache"])
        logits = torch.nn.functional.linear(h, w["codec_head.weight"]).float().view(rows, -1, cfg.codec_channels, cfg.codec_vocab_size)[:, -1]
        guided = OD.cfg_and_mask(cfg, logits, 3.0, False, 0.8)
        pred = OD.sample_next_token(guided.reshape(B * cfg.codec_channels, -1), 1.2, 0.95, 45, cfg.codec_eos_value)
        state.update(kv=kv, cache=c2, tok=pred.view(B, 1, cfg.codec_channels))
        return time.perf_counter() - t0

    times, probe = [], {}
    with torch.no_grad():
        one_step()                                       # warm-up: thread pool, page faults of the 14 GB of weights
        one_step()
        # thread count: one per physical core is SURVEY 8d's rule, but a 16-row step on 128 cores of two sockets is dominated by the
        # fork/join of many small ops -- time one step at a few counts and keep the fastest for the sample
        top = max(1, min(n_phys, usable))
        for nt in sorted({top, max(1, top // 2), max(1, top // 4), min(top, 16), min(top, 8)}, reverse=True):
            torch.set_num_threads(nt)
            one_step()                                   # (settle the pool at this size)
            probe[nt] = round(one_step() * 1e3, 1)
        torch.set_num_threads(min(probe, key=probe.get))
        for s in range(n_steps):
            times.append(one_step())
    times = [0.0, 0.0] + times
    steady = sorted(times[2:])                           # first two steps = warm-up (thread pool, page faults of the 14 GB of weights)
    sec = steady[len(steady) // 2]                       # median
    return {"value": round(B / sec, 3), "unit": "audio-tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model, "physical_cores": n_phys, "logical_cpus_usable": usable, "ms_per_step_by_threads": probe,
            "router": "C oracle (oracle/router_oracle.c through liboracle_router.so), not the reference's per-k Python loop: the port is "
                      "not op-for-op the reference, its routing is faster than the reference's",
            "sample": f"median of {len(steady)} decode steps (after 2 warm-up steps and a probe of the thread count: the fastest of {sorted(probe)} is used) of the full {cfg.num_hidden_layers}-layer model, batch {B} "
                      f"({2 * B} CFG rows), synthetic KV cache of {L} tokens, torch-CPU bf16 oracle (oracle/decode.py), "
                      f"{sec * 1e3:.0f} ms/step"}


def host_cpu():
    """(model string, physical cores, logical cpus usable by this process) of the host, from /proc/cpuinfo and the affinity mask."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None:
                cores.add((phys, core))
                phys = core = None
        if phys is not None:
            cores.add((phys, core))
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    n_phys = len(cores) or usable
    return model, n_phys, usable


def child_leg(script, argv, env=None, timeout=420):
    """Runs one of scripts/*_bench.py as a CHILD process (its own model and GPU context, after this process freed its memory) and
    returns the JSON object of its last output line; an error object if it fails -- the headline line must not be lost to a leg."""
    import subprocess
    e = dict(os.environ)
    e.update(env or {})
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script)] + argv, env=e, capture_output=True, text=True, timeout=timeout)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
        return json.loads(lines[-1])
    except Exception as ex:
        return {"error": repr(ex)}


def free_port() -> int:
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: start the N ranks (one per GPU) BEFORE anything here touches the GPU, as child
        # processes of this one (a process that initialised the GPU must never be replaced by another program)
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # the contract is ONE JSON line on stdout: native libraries (gloo's "[Gloo] Rank 0 is connected ..." banner) write to fd 1 too, so in a
    # multi-rank run everything written to fd 1 goes to stderr and the line goes to the saved descriptor
    line_fd = 1
    if world > 1:
        sys.stdout.flush()
        line_fd = os.dup(1)
        os.dup2(2, 1)
    ndev = torch.cuda.device_count()                   # (does not initialise the GPU)
    if ndev < 1:
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    shared_gpu = world > ndev                          # rehearsal of N ranks on fewer GPUs: ranks share cards, no RCCL
    torch.cuda.set_device(local % ndev)
    if shared_gpu:
        # workgroups of one launch hand rows to each other: they must all be resident, which processes sharing a card can only promise
        # each other when every rank's launches are sized to a SHARE of the compute units (the engine's co-residency guards then pick
        # the launch-per-kernel forms where a one-launch form would not fit its share)
        cus = torch.cuda.get_device_properties(local % ndev).multi_processor_count
        os.environ["UMOE_FAKE_CUS"] = str(max((cus - 16) // (world // ndev + (1 if world % ndev else 0)), 1))
    device = torch.device("cuda", local % ndev)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    import torch.distributed as dist
    rccl_ranks = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # control plane (barriers, MAX of the wall time, IPC handles) on gloo; RCCL ("nccl" on ROCm) for device collectives
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("cpu:gloo,cuda:nccl", device_id=device)
            try:
                probe = torch.ones(1, device=device)
                dist.all_reduce(probe)                   # RCCL over xGMI is up: every rank contributed
                torch.cuda.synchronize()
                rccl_ranks = int(probe.item())
            except Exception as e:
                print(f"[bench] rank {rank}: RCCL all-reduce failed ({e!r})", file=sys.stderr)
    mode = "single" if world == 1 else ("replica" if args.parallel == "replica" else "ep")
    cfg = make_cfg(args)
    B, K, W = args.batch, args.steps, args.warmup
    model, t_build = build_model(cfg, device)
    notes = {}
    ep_info = None
    if args.ep_trial:          # the child of the sandbox above: one short expert-parallel run, exit code = verdict
        from unimoe_audio_amd.ep import EpLink
        code = 0
        try:
            link = EpLink.from_dist(args.ep_trial, None, device)
            trial = decode_leg(model, cfg, args, device, rank, B, ep=link, profile=False, steps=4, warmup=2)
            code = 3 if trial["ep_error"] else 0
        except Exception as e:
            print(f"[bench ep-trial] rank {rank}: {e!r}", file=sys.stderr)
            code = 2
        dist.barrier()
        dist.destroy_process_group()
        raise SystemExit(code)
    if mode == "ep":
        from unimoe_audio_amd.ep import EpLink
        backends = [args.ep_backend] + [b for b in ("peer", "rccl") if b != args.ep_backend and not (b == "rccl" and shared_gpu)]
        for be in backends:
            # a backend that cannot be set up, or whose exchange times out in a short rehearsal, is skipped by ALL ranks together.
            # The FIRST rehearsal of a backend runs in a child process per rank (a 4-layer model, 4 steps, its own rendezvous port): an
            # exchange that faults on this node's links (nothing here has crossed a real xGMI link yet) takes down the child, not the run
            # -- the replicas' number still gets printed
            ok, why = 1, ""
            link = None
            if not args.ep_trial:
                import subprocess
                # (under torchrun the workers are CLIENTS of the agent's store at MASTER_PORT; the children rendezvous on a port of their
                #  own, so their rank 0 has to serve the store itself)
                env = dict(os.environ, MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29511")) + 101 + backends.index(be)),
                           TORCHELASTIC_USE_AGENT_STORE="False")
                try:
                    rr = subprocess.run([sys.executable, os.path.abspath(__file__), "--ep-trial", be, "--gpus", str(world), "--layers", "4",
                                         "--batch", str(args.batch), "--prompt", str(args.prompt)], env=env, timeout=300,
                                        stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
                    if rr.returncode != 0:
                        ok, why = 0, f"sandboxed rehearsal exited with {rr.returncode}: {rr.stderr.strip().splitlines()[-1][:200] if rr.stderr.strip() else ''}"
                except subprocess.TimeoutExpired:
                    ok, why = 0, "sandboxed rehearsal timed out"
                if -max_over_ranks(-float(ok)) < 1.0:
                    notes[f"ep_backend_{be}"] = f"not used: {why or 'the sandboxed rehearsal failed on another rank'}"
                    if rank == 0:
                        print(f"[bench] expert-parallel backend {be!r} not usable: {why or 'sandboxed rehearsal failed on another rank'}", file=sys.stderr)
                    continue
            try:
                link = EpLink.from_dist(be, None, device)
                trial = decode_leg(model, cfg, args, device, rank, B, ep=link, profile=False, steps=4, warmup=2)
                if trial["ep_error"]:
                    ok, why = 0, "receive timed out"
            except Exception as e:
                ok, why = 0, repr(e)
            ok_all = -max_over_ranks(-float(ok))
            if ok_all >= 1.0:
                ep_info = decode_leg(model, cfg, args, device, rank, B, ep=link)
                ep_info["backend"] = be
                if max_over_ranks(float(ep_info["ep_error"])) == 0.0:
                    break
                why = "receive timed out in the timed run"
                ep_info = None
            notes[f"ep_backend_{be}"] = f"not used: {why or 'another rank failed'}"
            if rank == 0:
                print(f"[bench] expert-parallel backend {be!r} not usable: {why or 'another rank failed'}", file=sys.stderr)
        if ep_info is None:
            mode = "replica"
            notes["ep"] = "no expert-parallel backend worked on this node: `value` is the replica number"
    rep_info = None
    if mode != "ep" or not args.no_replica_check:
        rep_info = decode_leg(model, cfg, args, device, rank, B)
    if mode == "ep" and rep_info is not None:
        # both layouts were timed with the same K steps and give the same tokens (checked below): `auto` reports the faster one as `value`
        # and names the other in config; --parallel ep keeps the expert-parallel number whatever the replicas do
        t_ep, t_rep = max_over_ranks(ep_info["dt"]), max_over_ranks(rep_info["dt"])
        same = float(torch.equal(ep_info["tokens"], rep_info["tokens"]))
        notes["ep_tokens_equal_replica"] = bool(-max_over_ranks(-same) >= 1.0)
        notes["ep_value"] = round(world * B * K / t_ep, 2)
        notes["ep_ms_per_step"] = round(t_ep / K * 1e3, 4)
        notes["ep_backend_used"] = ep_info["backend"]
        notes["replica_value"] = round(world * B * K / t_rep, 2)
        notes["replica_ms_per_step"] = round(t_rep / K * 1e3, 4)
        if args.parallel == "auto" and t_rep < t_ep:
            mode = "replica"
            notes["parallelism_choice"] = "auto: the N independent replicas were faster than the expert-parallel layout on this node; both numbers above"
        elif args.parallel == "auto":
            notes["parallelism_choice"] = "auto: the expert-parallel layout was faster than N independent replicas on this node; both numbers above"
    info = ep_info if mode == "ep" else rep_info
    dt = max_over_ranks(info["dt"])                                     # the slowest rank defines the step time
    out = None
    value = world * B * K / dt
    if rank == 0:
        ep = world if mode == "ep" else 1
        info_dt = dict(info)
        info_dt["dt"] = dt
        out = {
            "metric": "audio-tokens/sec/node (TTS decode, bs=8 per GPU)", "value": round(value, 2), "unit": "audio-tokens/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(dt / K * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: TTS decode batch 8 (16 CFG rows) per GPU, 300-token prompt, "
                                   f"{cfg.num_hidden_layers}-layer DCMoE (8+1 routed, 2 shared), 12x1027 codec head, "
                                   "CFG 3.0 + top-k 45 + top-p 0.95 sampling, 16 kHz DAC 50 frames/s"
                                   + ("; N GPUs = BASELINE configs[3] layout (rows batch-sharded, routed experts sharded n_real/N per GPU)"
                                      if mode == "ep" else ""),
                       "global_batch": world * B, "prompt_len": args.prompt, "kv_len_end": info["kv_len_end"] + 4,
                       "parallelism": {"single": "single", "ep": f"ep{world}", "replica": f"replica{world}"}[mode],
                       "graph": not args.no_graph, "codes_per_s": round(value * cfg.codec_channels, 1),
                       "prefill_s": round(info["t_prefill"], 3), "mean_top_k": round(info["mean_top_k"], 2)},
            "roofline": roofline(cfg, info_dt, args, ep),
            "kernel_ms_per_step": {k: round(v[0] * v[1], 4) for k, v in info["prof"].items()},
        }
        if world > 1:
            out["config"]["rccl_ranks"] = rccl_ranks
            out["config"]["ranks_share_gpus"] = shared_gpu
            if mode == "ep":
                out["config"]["ep_exchange"] = {"peer": "xGMI peer stores in the step graph (HIP IPC regions)",
                                                "rccl": "ncclAllGather + grouped ncclSend/ncclRecv captured in the step graph"}[info["backend"]]
            out["config"].update(notes)
    if world == 1 and args.ep_emulate > 1:
        from unimoe_audio_amd.ep import EpLink
        emu = decode_leg(model, cfg, args, device, 0, B, ep=EpLink(0, args.ep_emulate, "loopback"))
        out["ep_emulation"] = {"ep_size": args.ep_emulate, "ms_per_step": round(emu["dt"] / K * 1e3, 4),
                               "audio_tokens_per_s_per_gpu": round(B * K / emu["dt"], 2), "ep_error": emu["ep_error"],
                               "kernel_ms_per_step": {k: round(v[0] * v[1], 4) for k, v in emu["prof"].items()},
                               "expert_launch": emu.get("expert_launch"),
                               "roofline": {k: v for k, v in roofline(cfg, dict(emu, backend="loopback"), args, args.ep_emulate).items()
                                            if k in ("kernel", "achieved", "frac", "bytes_per_launch", "avg_launch_us", "step_bytes", "step_achieved", "step_frac")},
                               "note": "ONE rank of an expert-parallel job in loopback on one GPU (its own rows stand in for the "
                                       "peers'): kernel time of the sharded step without xGMI latency; not the model's outputs"}
    if world == 1 and not args.no_config1:
        # BASELINE configs[0]: the reference's own CPU-runnable case, one prompt (2 CFG rows: ragged dispatch path), GPU side
        try:
            c1 = decode_leg(model, cfg, args, device, 0, 1, profile=False, steps=min(K, 200), warmup=W)
            out["config1"] = {"workload": "BASELINE configs[0]: single prompt (batch 1 = 2 CFG rows), same model and sampling",
                              "value": round(c1["steps"] / c1["dt"], 2), "unit": "audio-tokens/s",
                              "ms_per_step": round(c1["dt"] / c1["steps"] * 1e3, 4), "steps": c1["steps"]}
        except Exception as e:
            out["config1"] = {"error": repr(e)}
    del model
    torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.layers and not args.codec_channels:
        # BASELINE configs[2] and configs[4] at their own shapes on this GPU (secondary keys; `value` stays configs[1])
        if not args.no_config3:
            out["config3"] = child_leg("train_bench.py", [], {"TB_STEPS": "4", "TB_NOCACHE_STEPS": "2"})
        if not args.no_config5:
            out["config5"] = child_leg("video_bench.py", ["--steps", "100", "--warmup", "5"])
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args)
                out["config"]["gpu_vs_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
                if "config1" in out and "value" in out["config1"]:
                    out["config1"]["cpu_baseline"] = cpu_baseline(args, batch=1, steps=3)
            except Exception as e:  # the GPU number must not be lost to a host-side problem
                out["cpu_baseline"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
