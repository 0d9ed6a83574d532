"""Measurement (not a test): the DAC side of a request at the reference's 16 kHz / 50 Hz geometry (descript-audio-codec 1.0.0: encoder_dim
64, decoder_dim 1536, rates 2 4 5 8, 12 codebooks), synthetic weights, one MI355X: decode of N frames of codes (RVQ from_codes + the
conv decoder) and encode of a prompt clip (conv encoder + 12-level residual RVQ).  Prints one JSON line.

  python scripts/dac_bench.py [--frames 500] [--prompt-seconds 4.5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from unimoe_audio_amd import dac as D


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=500)             # 10 s of 50 Hz frames
    ap.add_argument("--prompt-seconds", type=float, default=4.5)   # a 4.5 s reference clip
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = D.DacModel().init_random(5).to(dev).float()
    n_par = sum(p.numel() for p in m.parameters())
    codes = torch.randint(0, 1024, (1, 12, a.frames), device=dev)

    def t_ms(fn, n=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2] * 1e3

    with torch.no_grad():
        dec_ms = t_ms(lambda: m.decode(m.from_codes(codes)))
        wav = torch.randn(1, 1, int(a.prompt_seconds * 16000), device=dev) * 0.1
        enc_ms = t_ms(lambda: m.encode(m.preprocess(wav, 16000)))
    print(json.dumps({"workload": f"DAC 16 kHz / 50 Hz, {n_par / 1e6:.1f} M parameters, fp32 conv kernels, synthetic weights, 1 x MI355X",
                      "decode_frames": a.frames, "decode_audio_seconds": a.frames / 50.0, "decode_ms": round(dec_ms, 2),
                      "encode_prompt_seconds": a.prompt_seconds, "encode_ms": round(enc_ms, 2)}))


if __name__ == "__main__":
    main()
