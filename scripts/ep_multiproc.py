"""Expert-parallel decode with N PROCESSES (one engine each, exchange regions mapped through HIP IPC, control plane on gloo)
against an ep_size = 1 engine on the same prompts in every process: logits and tokens must be bit-identical.  With fewer GPUs
than ranks the ranks share cards (the IPC mapping, the flags and the graph are the same as across GPUs).
usage: python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 scripts/ep_multiproc.py [layers=2] [steps=6]"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_gpu_engine import build, prompt, small_cfg
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.ep import EpLink
    from unimoe_audio_amd.model import DecodeEngine
    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    backend = sys.argv[3] if len(sys.argv) > 3 else "peer"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    if world > ndev:
        # processes sharing a card: in-launch hand-offs need every workgroup of a launch resident, so every rank's engine is told it
        # owns a share of the compute units (its one-launch MoE half is then that many workgroups; see bench.py)
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        os.environ["UMOE_FAKE_CUS"] = str((cus - 16) // world)
    cfg = small_cfg(hidden_size=2048, num_attention_heads=16, num_key_value_heads=2, dynamic_intermediate_size=2752,
                    shared_intermediate_size=1376, num_hidden_layers=layers)
    m, _ = build(cfg, 1, 0.02)                       # same seed in every process: slices of one model
    gm = m.to(dev)
    B, T, MAXT = 8, 12, steps + 40
    C, V = cfg.codec_channels, cfg.codec_vocab_size
    ids, am, codec = prompt(cfg, B, T, 10 + rank, [3, 0, 1, 0] + [0] * 12)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)

    def start(eng):
        x = gm.calculate_input_embedding(ids.to(dev), codec.to(dev))
        eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev))
        eng.start_decode(pre, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True,
                         seed=77 + rank)

    ref = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64)
    start(ref)
    ref_logits = []
    for s in range(steps):
        ref.step(use_graph=False)
        ref_logits.append(ref.copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu())
    ref_tokens = ref.tokens.cpu().clone()
    ref.close()
    link = EpLink.from_dist(backend, None, dev)
    eng = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64, ep=link)
    start(eng)
    ok = True
    for s in range(steps):
        eng.step(use_graph=(s >= 2))                 # two eager steps, then the captured graph
        got = eng.copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu()
        if not torch.equal(got, ref_logits[s]):
            print(f"rank {rank} step {s}: logits differ, max abs {(got - ref_logits[s]).abs().max().item()}", flush=True)
            ok = False
    err = eng.ep_error()
    tk = torch.equal(eng.tokens.cpu(), ref_tokens)
    ok = ok and err == 0 and tk
    res = [None] * world
    dist.all_gather_object(res, (rank, ok, err, tk))
    dist.barrier()
    eng.close()
    if rank == 0:
        for r in res:
            print(f"rank {r[0]}: ok {r[1]} ep_error {r[2]} tokens identical {r[3]}")
        print(f"{world} processes on {ndev} GPU(s), backend {backend}, {layers} layers, {steps} steps: "
              f"{'BIT-IDENTICAL' if all(r[1] for r in res) else 'MISMATCH'}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if all(r[1] for r in res) else 1)


if __name__ == "__main__":
    main()
