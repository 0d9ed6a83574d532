"""DAC (descript-audio-codec 1.0.0, 16 kHz / 50 Hz variant) on the HIP path: waveform <-> RVQ codes.

Mirrors what the reference needs from the third-party package (reference utils/UniMoE_Audio_utils.py:56-134):
    Dac()                      weight discovery like the reference (DAC_WEIGHTS, <pkg>/dac_model/weights_16khz.pth, ...; no download: offline)
    Dac.encode(audio_path)     mono mix (:97-98), resample to 16 kHz (:101-110), preprocess, encoder + RVQ -> codes [T][12] as lists (:112-119)
    Dac.decode(codes, path, min_duration)   quantizer.from_codes -> decoder -> zero-pad to min_duration -> 16-bit PCM wav (:121-134)
`DacModel` keeps the package's module tree and parameter names (encoder.block.N..., quantizer.quantizers.N.{in_proj,out_proj,codebook},
decoder.model.N..., weight-norm pairs weight_g / weight_v, Snake alpha), so the published `weights_16khz.pth` loads by name; every
dimension comes from the checkpoint's metadata or tensor shapes.  The arithmetic runs in libumoe_hip.so (umoe_dac_conv1d,
umoe_dac_conv_transpose1d with the preceding Snake fused, umoe_rvq_nearest, umoe_rvq_from_codes).  PARITY UNPINNED: `dac`,
`audiotools` and `torchaudio` are absent offline, so the layers, the resampler (torchaudio's windowed-sinc formula) and the PCM
conversion are restated from their published definitions; tests compare the kernels with plain fp32 torch restatements.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import wave
from typing import List, Optional, Sequence

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev_f32(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
        raise L.UmoeError("DAC ops need contiguous float32 device tensors (there is no CPU path in the product)")
    return C.c_void_p(t.data_ptr())


def conv1d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, stride=1, dilation=1, padding=0,
           snake_alpha: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, tanh: bool = False) -> torch.Tensor:
    """y = [tanh](conv1d(snake?(x)) + bias) [+ resid]; x [B, Cin, L], w [Cout, Cin, K] (weight norm already folded)."""
    B, Cin, Lx = x.shape
    Cout, Cin2, K = w.shape
    assert Cin2 == Cin
    Lout = (Lx + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    y = torch.empty((B, Cout, Lout), dtype=torch.float32, device=x.device)
    if resid is not None:
        assert tuple(resid.shape) == tuple(y.shape)
    lo = C.c_int()
    L.check(L.lib().umoe_dac_conv1d(_dev_f32(x), _dev_f32(w), _dev_f32(bias), _dev_f32(snake_alpha), _dev_f32(resid), B, Cin, Lx, Cout, K,
                                    stride, dilation, padding, int(tanh), _dev_f32(y), C.byref(lo), _stream()), "umoe_dac_conv1d")
    assert lo.value == Lout
    return y


def conv_transpose1d(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], *, stride, padding, output_padding=0,
                     snake_alpha: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = conv_transpose1d(snake?(x)) + bias; x [B, Cin, L], w [Cin, Cout, K]."""
    B, Cin, Lx = x.shape
    Cin2, Cout, K = w.shape
    assert Cin2 == Cin
    Lout = (Lx - 1) * stride - 2 * padding + K + output_padding
    y = torch.empty((B, Cout, Lout), dtype=torch.float32, device=x.device)
    lo = C.c_int()
    L.check(L.lib().umoe_dac_conv_transpose1d(_dev_f32(x), _dev_f32(w), _dev_f32(bias), _dev_f32(snake_alpha), B, Cin, Lx, Cout, K, stride,
                                              padding, output_padding, _dev_f32(y), C.byref(lo), _stream()), "umoe_dac_conv_transpose1d")
    assert lo.value == Lout
    return y


# ----------------------------------------------------------------------------------------------- parameter containers
class Snake1d(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, channels, 1))


class _WN(nn.Module):
    """A weight-normalised conv as the package stores it (torch.nn.utils.weight_norm: weight_g, weight_v, bias)."""

    def __init__(self, shape, g_shape, bias_n, **geom):
        super().__init__()
        self.weight_g = nn.Parameter(torch.ones(g_shape))
        self.weight_v = nn.Parameter(torch.zeros(shape))
        self.bias = nn.Parameter(torch.zeros(bias_n))
        self.geom = geom

    def folded(self) -> torch.Tensor:
        v = self.weight_v.data.float()
        n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, 1, 1)
        return (self.weight_g.data.float() * v / n).contiguous()


def WNConv1d(cin, cout, kernel_size, stride=1, dilation=1, padding=0):
    return _WN((cout, cin, kernel_size), (cout, 1, 1), cout, stride=stride, dilation=dilation, padding=padding, transposed=False)


def WNConvTranspose1d(cin, cout, kernel_size, stride=1, padding=0, output_padding=0):
    return _WN((cin, cout, kernel_size), (cin, 1, 1), cout, stride=stride, padding=padding, output_padding=output_padding, transposed=True)


class _Seq(nn.Module):
    """nn.Sequential-like numbering under the attribute name the package uses (`block` / `model`)."""

    def __init__(self, name: str, mods: Sequence[nn.Module]):
        super().__init__()
        setattr(self, name, nn.ModuleList(mods))
        self._name = name

    def items(self):
        return list(getattr(self, self._name))


def ResidualUnit(dim, dilation):
    pad = ((7 - 1) * dilation) // 2
    return _Seq("block", [Snake1d(dim), WNConv1d(dim, dim, 7, dilation=dilation, padding=pad), Snake1d(dim), WNConv1d(dim, dim, 1)])


def EncoderBlock(dim, stride):
    return _Seq("block", [ResidualUnit(dim // 2, 1), ResidualUnit(dim // 2, 3), ResidualUnit(dim // 2, 9), Snake1d(dim // 2),
                          WNConv1d(dim // 2, dim, 2 * stride, stride=stride, padding=math.ceil(stride / 2))])


def DecoderBlock(cin, cout, stride, output_padding=0):
    return _Seq("block", [Snake1d(cin), WNConvTranspose1d(cin, cout, 2 * stride, stride=stride, padding=math.ceil(stride / 2),
                                                           output_padding=output_padding),
                          ResidualUnit(cout, 1), ResidualUnit(cout, 3), ResidualUnit(cout, 9)])


class VectorQuantize(nn.Module):
    def __init__(self, input_dim, codebook_size, codebook_dim):
        super().__init__()
        self.in_proj = WNConv1d(input_dim, codebook_dim, 1)
        self.out_proj = WNConv1d(codebook_dim, input_dim, 1)
        self.codebook = nn.Embedding(codebook_size, codebook_dim)


class ResidualVectorQuantize(nn.Module):
    def __init__(self, input_dim, n_codebooks, codebook_size, codebook_dim):
        super().__init__()
        self.quantizers = nn.ModuleList([VectorQuantize(input_dim, codebook_size, codebook_dim) for _ in range(n_codebooks)])


DAC_16KHZ = dict(encoder_dim=64, encoder_rates=[2, 4, 5, 8], decoder_dim=1536, decoder_rates=[8, 5, 4, 2], n_codebooks=12, codebook_size=1024,
                 codebook_dim=8, sample_rate=16000)


class DacModel(nn.Module):
    """dac.DAC's module tree; encode / decode on the HIP kernels."""

    def __init__(self, encoder_dim=64, encoder_rates=(2, 4, 5, 8), latent_dim=None, decoder_dim=1536, decoder_rates=(8, 5, 4, 2), n_codebooks=12,
                 codebook_size=1024, codebook_dim=8, sample_rate=16000, decoder_output_padding=False, **unused):
        super().__init__()
        self.sample_rate, self.hop_length = int(sample_rate), int(math.prod(encoder_rates))
        self.n_codebooks, self.codebook_size, self.codebook_dim = int(n_codebooks), int(codebook_size), int(codebook_dim)
        d = encoder_dim
        enc = [WNConv1d(1, d, 7, padding=3)]
        for s in encoder_rates:
            d *= 2
            enc.append(EncoderBlock(d, s))
        self.latent_dim = int(latent_dim) if latent_dim else d
        enc += [Snake1d(d), WNConv1d(d, self.latent_dim, 3, padding=1)]
        self.encoder = _Seq("block", enc)
        self.quantizer = ResidualVectorQuantize(self.latent_dim, n_codebooks, codebook_size, codebook_dim)
        dec = [WNConv1d(self.latent_dim, decoder_dim, 7, padding=3)]
        cout = decoder_dim
        for i, s in enumerate(decoder_rates):
            cin, cout = decoder_dim // 2 ** i, decoder_dim // 2 ** (i + 1)
            # (1.0.0 passes no output_padding: an odd rate makes the waveform a few samples shorter than frames * hop)
            dec.append(DecoderBlock(cin, cout, s, output_padding=(s % 2 if decoder_output_padding else 0)))
        dec += [Snake1d(cout), WNConv1d(cout, 1, 7, padding=3), nn.Tanh()]
        self.decoder = _Seq("model", dec)
        self._fold = None

    # ---- weights ------------------------------------------------------------------------------------------------------
    @classmethod
    def load(cls, path: str, device="cuda") -> "DacModel":
        """The package's checkpoint format (audiotools BaseModel.save): {"state_dict", "metadata": {"kwargs": ...}}; a bare
        state dict works too (16 kHz geometry unless the shapes say otherwise)."""
        obj = torch.load(path, map_location="cpu", weights_only=False)
        sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj
        kw = dict(DAC_16KHZ)
        # geometry from the tensors themselves (every DAC dimension is a load-time parameter), then the metadata where present
        if "encoder.block.0.weight_v" in sd and "decoder.model.0.weight_v" in sd:
            kw["encoder_dim"] = int(sd["encoder.block.0.weight_v"].shape[0])
            kw["decoder_dim"], kw["latent_dim"] = int(sd["decoder.model.0.weight_v"].shape[0]), int(sd["decoder.model.0.weight_v"].shape[1])
            er, dr, i = [], [], 1
            while f"encoder.block.{i}.block.4.weight_v" in sd:
                er.append(int(sd[f"encoder.block.{i}.block.4.weight_v"].shape[2]) // 2)
                i += 1
            i = 1
            while f"decoder.model.{i}.block.1.weight_v" in sd:
                dr.append(int(sd[f"decoder.model.{i}.block.1.weight_v"].shape[2]) // 2)
                i += 1
            if er and dr:
                kw["encoder_rates"], kw["decoder_rates"] = er, dr
        meta = obj.get("metadata", {}) if isinstance(obj, dict) else {}
        kw.update({k: v for k, v in meta.get("kwargs", {}).items() if k in kw or k == "latent_dim"})
        nq = 1 + max((int(k.split(".")[2]) for k in sd if k.startswith("quantizer.quantizers.")), default=kw["n_codebooks"] - 1)
        kw["n_codebooks"] = nq
        cbw = sd.get("quantizer.quantizers.0.codebook.weight")
        if cbw is not None:
            kw["codebook_size"], kw["codebook_dim"] = int(cbw.shape[0]), int(cbw.shape[1])
        m = cls(**kw)
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"DAC checkpoint {path!r} lacks {len(missing)} tensors, e.g. {missing[:4]}")
        return m.to(device).float().eval()

    @torch.no_grad()
    def init_random(self, seed: int = 0) -> "DacModel":
        g = torch.Generator().manual_seed(seed)
        for n, p in self.named_parameters():
            if n.endswith("alpha"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("weight_g"):
                p.copy_(0.5 + torch.rand(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.02 * torch.randn(p.shape, generator=g))
            elif n.endswith("codebook.weight"):
                p.copy_(torch.randn(p.shape, generator=g))
            else:
                fan = p[0].numel() if p.dim() > 1 else 1
                p.copy_(torch.randn(p.shape, generator=g) / fan ** 0.5)
        self._fold = None
        return self

    def _folded(self) -> dict:
        """weight norm folded once (g * v / |v|), keyed by module; the RVQ projections stacked for the RVQ kernels."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._fold is not None and self._fold["key"] == key:
            return self._fold
        f = {"key": key}
        for m in self.modules():
            if isinstance(m, _WN):
                f[m] = (m.folded(), m.bias.data.float().contiguous())
            elif isinstance(m, Snake1d):
                f[m] = m.alpha.data.float().reshape(-1).contiguous()
        q = self.quantizer.quantizers
        f["cb"] = torch.stack([v.codebook.weight.data.float() for v in q], 0).contiguous()                      # [NQ, CB, cd]
        f["in_w"] = torch.stack([f[v.in_proj][0][:, :, 0] for v in q], 0).contiguous()                           # [NQ, cd, Dl]
        f["in_b"] = torch.stack([f[v.in_proj][1] for v in q], 0).contiguous()
        f["out_w"] = torch.stack([f[v.out_proj][0][:, :, 0] for v in q], 0).contiguous()                         # [NQ, Dl, cd]
        f["out_b"] = torch.stack([f[v.out_proj][1] for v in q], 0).contiguous()
        self._fold = f
        return f

    # ---- graph walk: Snake is always followed by a conv, into whose input load it is fused -------------------------------
    def _run(self, mods, x, f):
        alpha = None
        for m in mods:
            if isinstance(m, Snake1d):
                alpha = f[m]
            elif isinstance(m, _WN):
                w, b = f[m]
                g = m.geom
                if g["transposed"]:
                    x = conv_transpose1d(x, w, b, stride=g["stride"], padding=g["padding"], output_padding=g["output_padding"], snake_alpha=alpha)
                else:
                    x = conv1d(x, w, b, stride=g["stride"], dilation=g["dilation"], padding=g["padding"], snake_alpha=alpha)
                alpha = None
            elif isinstance(m, _Seq) and len(m.items()) == 4 and isinstance(m.items()[0], Snake1d):      # ResidualUnit: x + block(x)
                a0, c0, a1, c1 = m.items()
                w0, b0 = f[c0]
                w1, b1 = f[c1]
                t = conv1d(x, w0, b0, dilation=c0.geom["dilation"], padding=c0.geom["padding"], snake_alpha=f[a0])
                x = conv1d(t, w1, b1, snake_alpha=f[a1], resid=x)
            elif isinstance(m, _Seq):
                x = self._run(m.items(), x, f)
            elif isinstance(m, nn.Tanh):
                raise AssertionError("tanh is fused into the last conv")
        return x

    def preprocess(self, audio: torch.Tensor, sample_rate: Optional[int] = None) -> torch.Tensor:
        """right-pad to a multiple of the hop length (dac.DAC.preprocess)"""
        assert sample_rate is None or sample_rate == self.sample_rate
        n = audio.shape[-1]
        pad = math.ceil(n / self.hop_length) * self.hop_length - n
        return torch.nn.functional.pad(audio, (0, pad))

    @torch.no_grad()
    def encode(self, x: torch.Tensor, n_quantizers: Optional[int] = None):
        """x [B, 1, L] float32 -> (z [B, Dl, T], codes [B, NQ, T] int64)"""
        f = self._folded()
        z = self._run(self.encoder.items(), x.float().contiguous(), f)
        nq = self.n_codebooks if n_quantizers is None else int(n_quantizers)
        codes = torch.stack([ops.rvq_nearest(z[b].contiguous(), f["cb"][:nq].contiguous(), f["in_w"][:nq].contiguous(), f["in_b"][:nq].contiguous(),
                                             f["out_w"][:nq].contiguous(), f["out_b"][:nq].contiguous()) for b in range(z.shape[0])], 0)
        return z, codes.long()

    @torch.no_grad()
    def from_codes(self, codes: torch.Tensor) -> torch.Tensor:
        """codes [B, NQ, T] -> z_q [B, Dl, T] (ResidualVectorQuantize.from_codes: sum of out_proj(codebook[code]))"""
        f = self._folded()
        nq = codes.shape[1]
        return torch.stack([ops.rvq_from_codes(codes[b].contiguous(), f["cb"][:nq].contiguous(), f["out_w"][:nq].contiguous(), f["out_b"][:nq].contiguous())
                            for b in range(codes.shape[0])], 0)

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """z [B, Dl, T] -> audio [B, 1, ~T * hop]"""
        f = self._folded()
        mods = self.decoder.items()
        x = self._run(mods[:-3], z.float().contiguous(), f)
        snake, last = mods[-3], mods[-2]
        w, b = f[last]
        return conv1d(x, w, b, padding=last.geom["padding"], snake_alpha=f[snake], tanh=True)


# ----------------------------------------------------------------------------------------------- resampler / wav io
def resample(wave_BxL: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.transforms.Resample's default algorithm (windowed sinc, Hann window), restated: a bank of `new/gcd` filters applied
    with stride `orig/gcd` (umoe_dac_resample)."""
    if orig_freq == new_freq:
        return wave_BxL
    g = math.gcd(int(orig_freq), int(new_freq))
    o, n = int(orig_freq) // g, int(new_freq) // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64)[:, None, None] / n + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)                 # [n, 1, 2*width + o]
    B, Lx = wave_BxL.shape
    Lout = math.ceil(n * Lx / o)
    x = wave_BxL.float().contiguous()
    kd = kern[:, 0].float().to(x.device).contiguous()                                                     # [n, 2*width + o]
    y = torch.empty((B, Lout), dtype=torch.float32, device=x.device)
    L.check(L.lib().umoe_dac_resample(_dev_f32(x), _dev_f32(kd), B, Lx, o, n, width, Lout, _dev_f32(y), _stream()), "umoe_dac_resample")
    return y


def read_wav(path: str):
    """PCM wav -> (float32 [channels, samples] in [-1, 1), sample_rate); stdlib only (audiotools / soundfile are absent offline)."""
    import numpy as np
    with wave.open(path, "rb") as wf:
        ch, sw, sr, n = wf.getnchannels(), wf.getsampwidth(), wf.getframerate(), wf.getnframes()
        raw = wf.readframes(n)
    if sw == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif sw == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif sw == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"unsupported sample width {sw} in {path}")
    return torch.from_numpy(a.reshape(-1, ch).T.copy()), sr


def write_wav_pcm16(path: str, audio_CxL: torch.Tensor, sample_rate: int):
    """torchaudio.save(..., encoding="PCM_S", bits_per_sample=16) (reference utils.py:134): scale, round, clamp, little endian."""
    a = (audio_CxL.detach().cpu().float() * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
    with wave.open(path, "wb") as wf:
        wf.setnchannels(a.shape[0])
        wf.setsampwidth(2)
        wf.setframerate(sample_rate)
        wf.writeframes(a.T.contiguous().numpy().tobytes())


class Dac:
    """reference utils/UniMoE_Audio_utils.py:56-134 (same methods and error behaviour; no download offline)."""

    def __init__(self, weights_path: Optional[str] = None, device="cuda", model: Optional[DacModel] = None):
        self.resampler = dict()
        if model is not None:
            self.model = model
            return
        base_dir = os.path.dirname(__file__)
        candidates = [p for p in (weights_path, os.environ.get("DAC_WEIGHTS")) if p]
        candidates.extend([os.path.join(base_dir, "dac_model", "weights_16khz.pth"), os.path.join(base_dir, "weights_16khz.pth"),
                           os.path.join(os.getcwd(), "utils", "dac_model", "weights_16khz.pth"),
                           os.path.join(os.getcwd(), "dac_model", "weights_16khz.pth")])
        final = next((p for p in candidates if p and os.path.isfile(p)), None)
        if not final:
            raise FileNotFoundError("DAC weights not found. Please place weights_16khz.pth in one of the following locations or set "
                                    "DAC_WEIGHTS to an absolute path:" + "\n - " + "\n - ".join(candidates))
        self.model = DacModel.load(final, device)

    @property
    def device(self):
        return next(self.model.parameters()).device

    def encode(self, audio_path) -> List[List[int]]:
        audio, sr = read_wav(audio_path)
        if audio.shape[0] == 2:
            audio = 0.5 * (audio[:1] + audio[1:])                                 # utils.py:97-98
        audio = audio.to(self.device)
        if sr != self.model.sample_rate:
            audio = resample(audio, sr, self.model.sample_rate)                   # utils.py:101-110
        x = self.model.preprocess(audio[None], self.model.sample_rate)
        _, codes = self.model.encode(x)
        codes = codes[0].transpose(0, 1)
        assert codes.shape[1] == self.model.n_codebooks and codes.dim() == 2      # utils.py:116
        return codes.tolist()

    def decode(self, codes: torch.Tensor, save_path: str, min_duration=None):
        assert codes.shape[0] == 1 and codes.shape[1] == self.model.n_codebooks   # utils.py:122
        z = self.model.from_codes(codes.to(self.device))
        audio_out = self.model.decode(z)[0].detach().cpu()
        sr = self.model.sample_rate
        duration = audio_out.size(1) / sr
        if min_duration is not None and duration < min_duration:                  # utils.py:127-132
            pad = torch.zeros((audio_out.size(0), int((min_duration - duration) * sr)), dtype=audio_out.dtype)
            audio_out = torch.cat((audio_out, pad), dim=1)
        write_wav_pcm16(save_path, audio_out, sr)
