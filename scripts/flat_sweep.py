"""A/B of decode-step variants in ONE process on ONE engine (cdna_hip_programming.md rule 24: interleaved rounds, one device).

  python scripts/flat_sweep.py [--layers 36] [--steps 300] [--rounds 3] NAME=K1:V1,K2:V2 ...

Every variant is a set of environment switches that the C library reads when the step graph is captured (UMOE_FLAT_MOE, the
UMOE_FLAT_* schedule-model constants, ...).  Each timing block restarts the decode at the prompt (same KV lengths for everybody),
re-captures the graph, runs `warm` untimed steps and times `steps` replays.  Prints one JSON line per variant: median / min ms per step
over the rounds, and whether its generated tokens equal the first variant's (they must: every variant is bit-identical by design)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=36)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warm", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    variants = []
    for v in a.variants or ["base="]:
        name, _, kv = v.partition("=")
        env = dict(x.split(":", 1) for x in kv.split(",") if x)
        variants.append((name, env))
    keys = sorted({k for _, e in variants for k in e})
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.model import DecodeEngine
    args = argparse.Namespace(layers=a.layers, codec_channels=0, prompt=300)
    cfg = bench.make_cfg(args)
    dev = torch.device("cuda:0")
    model, t_build = bench.build_model(cfg, dev)
    B, T = a.batch, 300
    max_tokens = a.steps + a.warm + 64
    eng = DecodeEngine(model, B, Lmax=T + max_tokens + 8, Tmax=max_tokens + 64)
    ids, am, codec = bench.synth_prompt(cfg, B, T, dev)
    x = model.calculate_input_embedding(ids, codec)
    eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    res = {n: [] for n, _ in variants}
    toks = {}
    for r in range(a.rounds):
        for name, env in variants:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env)
            eng.start_decode(pre, psteps, max_tokens, max_tokens, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True, seed=1234)
            for _ in range(a.warm):
                eng.step(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.step(True)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / a.steps * 1e3)
            code = eng.handoff_error()
            if code:
                print(json.dumps({"variant": name, "handoff_error": code}), flush=True)
                return 1
            if r == 0:
                toks[name] = eng.tokens[:, : a.steps + a.warm].cpu().clone()
    first = variants[0][0]
    for name, env in variants:
        v = sorted(res[name])
        print(json.dumps({"variant": name, "env": env, "ms_per_step_median": round(v[len(v) // 2], 4), "min": round(v[0], 4), "all": [round(t, 4) for t in res[name]],
                          "tokens_equal_first": bool(torch.equal(toks[name], toks[first])), "layers": a.layers}), flush=True)
    eng.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
