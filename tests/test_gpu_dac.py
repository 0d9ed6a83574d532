"""GPU tests of the DAC conv stacks (unimoe_audio_amd/dac.py over umoe_dac_conv1d / umoe_dac_conv_transpose1d / umoe_rvq_*).
PARITY UNPINNED: descript-audio-codec 1.0.0, audiotools and torchaudio are absent offline, so the checker is a plain
fp32 torch restatement of the same published layers on the CPU (floating-point kernels: tolerance 2e-4 relative to the tensor's
scale; code ids exact except where the top-2 similarity gap is below fp32 resolution)."""
import math
import os
import wave

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from unimoe_audio_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def snake(x, a):
    a = a.reshape(1, -1, 1)
    return x + torch.sin(a * x) ** 2 / (a + 1e-9)


def close(a, b, tol=2e-4):
    return float((a - b).abs().max()) <= tol * max(float(b.abs().max()), 1e-6)


@pytest.mark.parametrize("cin,cout,K,stride,dil,pad,use_snake,use_res,tanh", [
    (1, 64, 7, 1, 1, 3, False, False, False),        # encoder stem
    (64, 64, 7, 1, 9, 27, True, False, False),       # residual unit, dilation 9
    (64, 64, 1, 1, 1, 0, True, True, False),         # residual unit, 1x1 + skip
    (64, 128, 4, 2, 1, 1, True, False, False),       # encoder down-sampling, rate 2
    (96, 130, 10, 5, 1, 3, True, False, False),      # rate 5 (odd), ragged channel count
    (72, 64, 16, 8, 1, 4, True, False, False),       # rate 8
    (96, 1, 7, 1, 1, 3, True, False, True),          # decoder head: Snake -> conv -> tanh
])
def test_conv1d_vs_torch(dev, cin, cout, K, stride, dil, pad, use_snake, use_res, tanh):
    from unimoe_audio_amd import dac as D
    g = torch.Generator().manual_seed(cin * 131 + cout + K)
    B, L = 2, 333
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, K, generator=g) / (cin * K) ** 0.5
    b = 0.1 * torch.randn(cout, generator=g)
    a = 0.5 + torch.rand(cin, generator=g)
    ref = F.conv1d(snake(x, a) if use_snake else x, w, b, stride=stride, dilation=dil, padding=pad)
    res = torch.randn(ref.shape, generator=g) if use_res else None
    if tanh:
        ref = torch.tanh(ref)
    if use_res:
        ref = ref + res
    y = D.conv1d(x.to(dev), w.to(dev), b.to(dev), stride=stride, dilation=dil, padding=pad, snake_alpha=a.to(dev) if use_snake else None,
                 resid=None if res is None else res.to(dev), tanh=tanh).cpu()
    assert y.shape == ref.shape and close(y, ref)


@pytest.mark.parametrize("cin,cout,stride,out_pad", [(128, 64, 8, 0), (96, 48, 5, 0), (96, 48, 5, 1), (64, 40, 4, 0), (48, 24, 2, 0)])
def test_conv_transpose1d_vs_torch(dev, cin, cout, stride, out_pad):
    from unimoe_audio_amd import dac as D
    g = torch.Generator().manual_seed(cin + stride)
    B, L, K, pad = 2, 77, 2 * stride, math.ceil(stride / 2)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cin, cout, K, generator=g) / (cin * 2) ** 0.5
    b = 0.1 * torch.randn(cout, generator=g)
    a = 0.5 + torch.rand(cin, generator=g)
    ref = F.conv_transpose1d(snake(x, a), w, b, stride=stride, padding=pad, output_padding=out_pad)
    y = D.conv_transpose1d(x.to(dev), w.to(dev), b.to(dev), stride=stride, padding=pad, output_padding=out_pad, snake_alpha=a.to(dev)).cpu()
    assert y.shape == ref.shape and close(y, ref)


def torch_graph(m, f_cpu, x, mods):
    """plain torch restatement of DacModel._run on CPU tensors (weights = the folded tensors copied to the host)"""
    from unimoe_audio_amd import dac as D
    alpha = None
    for mod in mods:
        if isinstance(mod, D.Snake1d):
            alpha = f_cpu[mod]
        elif isinstance(mod, D._WN):
            w, b = f_cpu[mod]
            gm = mod.geom
            xin = snake(x, alpha) if alpha is not None else x
            if gm["transposed"]:
                x = F.conv_transpose1d(xin, w, b, stride=gm["stride"], padding=gm["padding"], output_padding=gm["output_padding"])
            else:
                x = F.conv1d(xin, w, b, stride=gm["stride"], dilation=gm["dilation"], padding=gm["padding"])
            alpha = None
        elif isinstance(mod, D._Seq) and len(mod.items()) == 4 and isinstance(mod.items()[0], D.Snake1d):
            y = torch_graph(m, f_cpu, x, mod.items())
            x = x + y
        elif isinstance(mod, D._Seq):
            x = torch_graph(m, f_cpu, x, mod.items())
        elif isinstance(mod, torch.nn.Tanh):
            x = torch.tanh(x)
    return x


def test_dac_model_encode_decode_vs_torch_restatement(dev):
    """The whole graph at the 16 kHz geometry (rates 2,4,5,8 / 8,5,4,2, 12 codebooks of 1024 x 8) with narrower channels."""
    from unimoe_audio_amd import dac as D
    m = D.DacModel(encoder_dim=16, encoder_rates=[2, 4, 5, 8], decoder_dim=192, decoder_rates=[8, 5, 4, 2], n_codebooks=12, codebook_size=1024,
                   codebook_dim=8).init_random(3)
    assert m.hop_length == 320 and m.latent_dim == 256
    with torch.no_grad():     # random weights: damp the residual branches (24 units, each would double the variance) and the decoder head so
        for n, p in m.named_parameters():      # that the waveform stays inside tanh's linear range and the comparison means something
            if n.endswith("block.3.weight_g"):
                p.mul_(0.15)
            if n == "decoder.model.6.weight_g":
                p.mul_(0.05)
    gm = m.to(dev).float()
    f = gm._folded()
    f_cpu = {k: (tuple(t.cpu() for t in v) if isinstance(v, tuple) else v.cpu()) for k, v in f.items() if not isinstance(k, str)}
    g = torch.Generator().manual_seed(4)
    audio = 0.3 * torch.randn(1, 1, 320 * 40 - 57, generator=g)
    x = gm.preprocess(audio)
    assert x.shape[-1] == 320 * 40
    z, codes = gm.encode(x.to(dev))
    z_ref = torch_graph(m, f_cpu, x, m.encoder.items())
    assert z.shape == z_ref.shape == (1, 256, 40) and close(z.cpu(), z_ref, 5e-4)
    # RVQ on the kernel's own z (fp64 restatement; near-ties excluded as in test_gpu_ops)
    res = z[0].cpu().double().t().clone()
    cb, in_w, in_b, out_w, out_b = (f[k].cpu().double() for k in ("cb", "in_w", "in_b", "out_w", "out_b"))
    for q in range(12):
        e = res @ in_w[q].t() + in_b[q]
        sim = F.normalize(e, dim=-1) @ F.normalize(cb[q], dim=-1).t()
        top2 = sim.topk(2, dim=-1)
        clear = (top2.values[:, 0] - top2.values[:, 1]) > 1e-5
        assert torch.equal(codes[0, q].cpu()[clear], top2.indices[:, 0][clear]), q
        res = res - cb[q][codes[0, q].cpu()] @ out_w[q].t() - out_b[q]
    zq = gm.from_codes(codes)
    assert close(zq[0].cpu().double().t(), z[0].cpu().double().t() - res, 1e-3)
    wav = gm.decode(zq)
    wav_ref = torch_graph(m, f_cpu, zq.cpu(), m.decoder.items())
    assert wav.shape == wav_ref.shape and wav.shape[:2] == (1, 1)
    assert wav.shape[-1] == 320 * 40 - 8                    # no output_padding in 1.0.0: the rate-5 stage loses one sample, x 4 x 2
    assert 0.01 < float(wav_ref.abs().max()) < 0.999, float(wav_ref.abs().max())       # neither dead nor saturated
    assert close(wav.cpu(), wav_ref, 1e-3)


def test_resample_vs_direct_sinc_sum(dev):
    """the restated torchaudio formula, evaluated sample by sample in numpy float64"""
    from unimoe_audio_amd import dac as D
    rng = np.random.default_rng(0)
    for orig, new in ((44100, 16000), (24000, 16000), (8000, 16000)):
        x = rng.standard_normal(700)
        y = D.resample(torch.tensor(x, dtype=torch.float32, device=dev)[None], orig, new)[0].cpu().numpy()
        gcd = math.gcd(orig, new)
        o, n = orig // gcd, new // gcd
        lpw, base = 6, min(o, n) * 0.99
        width = math.ceil(lpw * o / base)
        assert y.shape[0] == math.ceil(n * 700 / o)
        for j in list(range(0, 40)) + list(range(y.shape[0] - 40, y.shape[0])):
            fr, ph = divmod(j, n)
            idx = np.arange(-width, width + o) / o
            t = np.clip((-ph / n + idx) * base, -lpw, lpw)
            k = np.where(t == 0, 1.0, np.sin(t * np.pi) / np.where(t == 0, 1.0, t * np.pi)) * np.cos(t * np.pi / lpw / 2) ** 2 * (base / o)
            pos = fr * o + np.arange(-width, width + o)
            xv = np.where((pos >= 0) & (pos < 700), x[np.clip(pos, 0, 699)], 0.0)
            assert abs(float((k * xv).sum()) - float(y[j])) < 2e-4, (orig, new, j)


def test_dac_wrapper_wav_round_trip(dev, tmp_path):
    """Dac.encode / Dac.decode (reference utils.py:95-134): stereo 22.05 kHz wav in -> codes [T][12] as lists; codes -> 16-bit PCM
    16 kHz wav, zero-padded to min_duration."""
    from unimoe_audio_amd import dac as D
    m = D.DacModel(encoder_dim=16, decoder_dim=192).init_random(5).to(dev).float()
    codec = D.Dac(model=m)
    sr, n = 22050, 22050
    t = np.arange(n) / sr
    stereo = np.stack([0.4 * np.sin(2 * np.pi * 440 * t), 0.2 * np.sin(2 * np.pi * 660 * t)], 1)
    src = str(tmp_path / "in.wav")
    with wave.open(src, "wb") as wf:
        wf.setnchannels(2); wf.setsampwidth(2); wf.setframerate(sr)
        wf.writeframes((stereo * 32767).astype("<i2").tobytes())
    codes = codec.encode(src)
    T = math.ceil(16000 / 320)                       # 1 s -> 16 000 samples -> 50 frames
    assert isinstance(codes, list) and len(codes) == T and len(codes[0]) == 12 and all(0 <= c < 1024 for row in codes for c in row)
    out = str(tmp_path / "out.wav")
    codec.decode(torch.tensor(codes).t()[None], out, min_duration=1.5)
    with wave.open(out, "rb") as wf:
        assert wf.getframerate() == 16000 and wf.getsampwidth() == 2 and wf.getnchannels() == 1
        assert wf.getnframes() == T * 320 - 8 + int((1.5 - (T * 320 - 8) / 16000) * 16000)
    with pytest.raises(AssertionError):
        codec.decode(torch.zeros(1, 11, 5, dtype=torch.long), out)
