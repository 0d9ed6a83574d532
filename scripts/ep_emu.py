"""ONE rank of an expert-parallel decode job in loopback on one GPU (no peers): the kernel-time view of the sharded step.
usage: python scripts/ep_emu.py [ep=8] [steps=50] [layers=36]      (run under rocprofv3 --kernel-trace for per-kernel times)"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ep = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    layers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    args = types.SimpleNamespace(layers=layers, codec_channels=0, prompt=300, steps=steps, warmup=5, batch=8, no_graph=False)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = bench.make_cfg(args)
    model, _ = bench.build_model(cfg, dev)
    from unimoe_audio_amd.ep import EpLink
    link = EpLink(0, ep, "loopback") if ep > 1 else None
    info = bench.decode_leg(model, cfg, args, dev, 0, args.batch, ep=link, profile=True)
    print(f"ep {ep}: {info['dt'] / steps * 1e3:.4f} ms/step, ep_error {info['ep_error']}")
    print({k: round(v[0] * 1e3, 2) for k, v in info["prof"].items()})


if __name__ == "__main__":
    main()
