"""Expert-parallel decode with N virtual ranks on ONE GPU (one process, one engine per rank, regions connected directly)
against N independent ep_size = 1 engines on the same prompts: logits and sampled tokens must be bit-identical.
usage: python scripts/ep_virtual.py [ranks=2] [layers=2] [steps=6] [graph=0/1]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from test_gpu_engine import build, prompt, small_cfg
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.ep import EpLink
    from unimoe_audio_amd.model import DecodeEngine
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    graph = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False
    dev = torch.device("cuda:0")
    cfg = small_cfg(hidden_size=2048, num_attention_heads=16, num_key_value_heads=2, dynamic_intermediate_size=2752,
                    shared_intermediate_size=1376, num_hidden_layers=layers)
    m, _ = build(cfg, 1, 0.02)
    gm = m.to(dev)
    B, T, MAXT = 8, 12, steps + 40
    C, V = cfg.codec_channels, cfg.codec_vocab_size
    prompts = [prompt(cfg, B, T, 10 + r, [3, 0, 1, 0] + [0] * 12) for r in range(N)]
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)

    def start(eng, r):
        ids, am, codec = prompts[r]
        x = gm.calculate_input_embedding(ids.to(dev), codec.to(dev))
        eng.prefill(x.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev))
        eng.start_decode(pre, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.2, top_p=0.95, top_k=45, eos_mul=0.8, do_sample=True,
                         seed=77 + r)

    ref_logits, ref_tokens = [], []
    for r in range(N):
        eng = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64)
        start(eng, r)
        lg = []
        for s in range(steps):
            eng.step(use_graph=graph)
            lg.append(eng.copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu())
        ref_logits.append(lg)
        ref_tokens.append(eng.tokens.cpu().clone())
        eng.close()
    links = [EpLink(r, N, "peer") for r in range(N)]
    # N ranks share this ONE card: the one-launch MoE half (umoe_moe_ep.hip) needs every workgroup of every rank's launch resident at
    # once, so each rank's engine is told it owns 1/N of the compute units (its launch is then that many workgroups)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    # (16 CUs stay free: a rank's small launches -- attention, QKV -- must find room while the other ranks' expert launches spin)
    share = (cus - 16) // N
    os.environ["UMOE_FAKE_CUS"] = str(share)
    engs = [DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64, ep=links[r], ep_connect=False) for r in range(N)]
    EpLink.local_mesh([e.h for e in engs])
    del os.environ["UMOE_FAKE_CUS"]
    print(f"expert launch per rank: {'one launch, ' + str(share) + ' workgroups' if os.environ.get('UMOE_EP_FLAT', '1') != '0' else 'launch per kernel'}")
    streams = [torch.cuda.Stream() for _ in range(N)]
    for r in range(N):
        with torch.cuda.stream(streams[r]):
            start(engs[r], r)
    torch.cuda.synchronize()
    ok = True
    t0 = time.time()
    for s in range(steps):
        for r in range(N):
            with torch.cuda.stream(streams[r]):
                engs[r].step(use_graph=graph)
        torch.cuda.synchronize()
        for r in range(N):
            got = engs[r].copy_buffer("logits", torch.float32, (2 * B, C * V)).cpu()
            same = torch.equal(got, ref_logits[r][s])
            if not same:
                d = (got - ref_logits[r][s]).abs().max().item()
                print(f"step {s} rank {r}: logits differ, max abs {d}")
                ok = False
    for r in range(N):
        err = engs[r].ep_error()
        tk = torch.equal(engs[r].tokens.cpu(), ref_tokens[r])
        print(f"rank {r}: ep_error {err} tokens identical {tk}")
        ok = ok and err == 0 and tk
    print(f"{N} virtual ranks, {layers} layers, {steps} steps, graph={graph}: {'BIT-IDENTICAL' if ok else 'MISMATCH'} ({time.time() - t0:.2f} s)")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
