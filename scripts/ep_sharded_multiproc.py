"""Expert-parallel decode with N PROCESSES whose models hold ONLY their local experts (reference: num_local_experts = 8 // ep_size,
utils/UniMoE_Audio_core.py:505; per-rank expert files, deepspeed_ep_param_aggregation.py:16-48): the engine of every rank packs its
n_real / N experts only, the prompt runs through the module-level forward whose DCMoE blocks exchange the routed rows between the ranks
(core.py:455-488 over torch.distributed), the engine takes the KV cache over (umoe_engine_prefill_external) and decodes expert-parallel.
Checked against an ep_size 1 engine on the FULL model with the same prompts: logits of every step and all tokens bit-identical, and the
routed experts' bytes a rank holds = 1 / N of the full model's.
usage: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/ep_sharded_multiproc.py [layers=2] [steps=6]"""
import copy
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_gpu_engine import build, prompt, small_cfg
    from unimoe_audio_amd.checkpoint import ep_local_key
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.ep import EpLink
    from unimoe_audio_amd.model import DecodeEngine, UniAudioRVQQwen2_5VLMoEForConditionalGeneration as Model
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    if world > ndev:
        # ranks share a card: every rank's one-launch MoE half takes its share of the compute units (all its workgroups must be resident)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        os.environ["UMOE_FAKE_CUS"] = str((cus - 16) // world)
    cfg = small_cfg(hidden_size=2048, num_attention_heads=16, num_key_value_heads=2, dynamic_intermediate_size=2752,
                    shared_intermediate_size=1376, num_hidden_layers=layers)
    m, _ = build(cfg, 1, 0.02)                       # the FULL model, same seed in every process
    gm = m.to(dev)
    n_real = cfg.mlp_dynamic_expert_num
    B, T, MAXT = 8, 12, steps + 40
    C, V = cfg.codec_channels, cfg.codec_vocab_size
    ids, am, codec = prompt(cfg, B, T, 10 + rank, [3, 0, 1, 0] + [0] * 12)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)

    def start(eng, model, external):
        x = model.calculate_input_embedding(ids.to(dev), codec.to(dev))
        eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev), external=external)
        eng.start_decode(pre, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True,
                         seed=77 + rank)

    # reference: ep_size 1 on the full model, prompt through the same module-level forward; and through the engine's own prefill
    refs = {}
    for external in (True, False):
        ref = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64)
        start(ref, gm, external)
        lg = []
        for s in range(steps):
            ref.step(use_graph=False)
            lg.append(ref.copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu())
        refs[external] = (lg, ref.tokens.cpu().clone())
        ref.close()
    native_equal = all(torch.equal(a, b) for a, b in zip(refs[True][0], refs[False][0])) and torch.equal(refs[True][1], refs[False][1])
    ref_logits, ref_tokens = refs[True]
    # the sharded model: ep_size in its config -> n_real / world expert modules per layer, filled with this rank's experts
    cfg_s = copy.deepcopy(cfg)
    cfg_s.ep_size = world
    ms = Model(cfg_s)
    full = gm.state_dict()
    sd = {}
    for k, v in full.items():
        lk = ep_local_key(k, n_real, rank, world)
        if lk is not None:
            sd[lk] = v
    missing, unexpected = ms.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "visual" not in k] and not unexpected, (missing[:5], unexpected[:5])
    ms = ms.to(dev).to(torch.bfloat16).eval()
    for layer in ms.language_model.layers:
        layer.mlp.dynamic_real_moe.set_deepspeed_parallelism(ep_group=dist.group.WORLD)
    is_expert = lambda n: "deepspeed_experts" in n
    bytes_full = sum(p.numel() * p.element_size() for n, p in gm.named_parameters() if is_expert(n))
    bytes_here = sum(p.numel() * p.element_size() for n, p in ms.named_parameters() if is_expert(n))
    del gm, m, full, sd
    torch.cuda.empty_cache()
    link = EpLink.from_dist("peer", None, dev)
    eng = DecodeEngine(ms, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64, ep=link)
    assert eng.sharded
    start(eng, ms, False)
    ok = True
    for s in range(steps):
        eng.step(use_graph=(s >= 2))
        got = eng.copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu()
        if not torch.equal(got, ref_logits[s]):
            print(f"rank {rank} step {s}: logits differ, max abs {(got - ref_logits[s]).abs().max().item()}", flush=True)
            ok = False
    err = eng.ep_error()
    tk = torch.equal(eng.tokens.cpu(), ref_tokens)
    ok = ok and err == 0 and tk and bytes_here * world == bytes_full
    res = [None] * world
    dist.all_gather_object(res, (rank, ok, err, tk, bytes_here, bytes_full, native_equal, eng.info("expert_launch")))
    dist.barrier()
    eng.close()
    if rank == 0:
        for r in res:
            print(f"rank {r[0]}: ok {r[1]} ep_error {r[2]} tokens identical {r[3]} routed-expert bytes {r[4]} of {r[5]} (1/{r[5] // max(r[4], 1)}) "
                  f"module-level prefill == engine prefill at ep 1: {r[6]} expert launch {r[7]}")
        print(f"{world} processes on {ndev} GPU(s), sharded weights, {layers} layers, {steps} steps: "
              f"{'SHARDED-BIT-IDENTICAL' if all(r[1] for r in res) else 'MISMATCH'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if all(r[1] for r in res) else 1)


if __name__ == "__main__":
    main()
