"""Host-side integer helpers of the DAC-token stream: delay pattern, prompt preparation, DecoderOutput.

Mirrors the reference names and semantics (reference utils/UniMoE_Audio_utils.py:137-325 and
utils/UniMoE_Audio_mod.py:140-156).  Pure index arithmetic on small int tensors; runs wherever the tensors live.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def _shifted_time(T: int, delay: Sequence[int], sign: int, device=None) -> torch.Tensor:
    t = torch.arange(T, device=device).view(T, 1)
    d = torch.as_tensor(list(delay), device=device).view(1, -1)
    return t + sign * d                                           # [T, C]


def build_delay_indices(B: int, T: int, C: int, delay_pattern: List[int]):
    """reference utils.py:137-165: returns (t - delay_c, clamped gather time)."""
    t = _shifted_time(T, delay_pattern, -1)
    return t.unsqueeze(0).expand(B, T, C), t.clamp(0, T - 1).unsqueeze(0).expand(B, T, C)


def apply_audio_delay(audio_BxTxC: torch.Tensor, pad_value: int, bos_value: int, precomp=None,
                      delay_pattern: Optional[List[int]] = None) -> torch.Tensor:
    """out[b,t,c] = in[b,t-d_c,c]; BOS where t-d_c < 0; PAD where t-d_c >= T (reference utils.py:168-183)."""
    B, T, C = audio_BxTxC.shape
    if precomp is None:
        precomp = build_delay_indices(B, T, C, delay_pattern)
    t, tc = (p.to(audio_BxTxC.device) for p in precomp)
    g = torch.gather(audio_BxTxC, 1, tc)
    out = torch.where(t < 0, torch.full_like(g, bos_value), g)
    return torch.where(t >= T, torch.full_like(g, pad_value), out)


def build_revert_indices(B: int, T: int, C: int, delay_pattern: List[int]):
    """reference utils.py:186-206"""
    t = _shifted_time(T, delay_pattern, +1).clamp(max=T - 1)
    return t.unsqueeze(0).expand(B, T, C), t.unsqueeze(0).expand(B, T, C)


def revert_audio_delay(audio_BxTxC: torch.Tensor, pad_value: int, precomp=None, T: Optional[int] = None,
                       delay_pattern: Optional[List[int]] = None) -> torch.Tensor:
    """out[b,t,c] = in[b,min(t+d_c,T-1),c] (reference utils.py:209-227)."""
    B, T_, C = audio_BxTxC.shape
    if precomp is None:
        precomp = build_revert_indices(B, T_, C, delay_pattern)
    t, _ = (p.to(audio_BxTxC.device) for p in precomp)
    g = torch.gather(audio_BxTxC, 1, t)
    lim = T_ if T is None else T
    return torch.where(t >= lim, torch.full_like(g, pad_value), g)


def prepare_audio_prompt(config, audio_prompts: list, device="cpu") -> Tuple[torch.Tensor, List[int]]:
    """reference `_prepare_audio_prompt`, utils.py:230-268: [B, max_len + max_delay + 1, C] int32, -1 = to generate."""
    C, bos, delay = config.codec_channels, config.codec_bos_value, list(config.codec_delay_pattern)
    longest = max((p.shape[0] if p is not None else 0) for p in audio_prompts)
    T = longest + max(delay) + 1
    buf = torch.full((len(audio_prompts), T, C), -1, dtype=torch.int32, device=device)
    buf[:, 0] = bos
    steps = []
    for i, p in enumerate(audio_prompts):
        n = 0 if p is None else p.shape[0]
        if n:
            buf[i, 1:n + 1] = p.to(device=device, dtype=torch.int32)
        steps.append(n + 1)
    return apply_audio_delay(buf, -1, bos, delay_pattern=delay), steps


class DecoderOutput:
    """reference utils.py:271-298 (same constructor / accessors; generation itself updates a device buffer)."""

    def __init__(self, prefill, prefill_steps, device, labels_prefill=None):
        self.generated_tokens = prefill
        self.prefill_steps = prefill_steps
        self.labels_prefill = labels_prefill
        self.device = device

    def get_tokens_at(self, step_from: int, step_to: Optional[int] = None) -> torch.Tensor:
        step_to = step_from + 1 if step_to is None else step_to
        return self.generated_tokens[:, step_from:step_to, :].to(self.device)

    def get_labels_at(self, step_from: int, step_to: Optional[int] = None):
        if self.labels_prefill is None:
            return None
        step_to = step_from + 1 if step_to is None else step_to
        return self.labels_prefill[:, step_from:step_to, :].to(self.device)

    def update_one(self, dec_out: torch.Tensor, step: int, apply_mask: bool = False):
        dec_out = dec_out.to(self.generated_tokens.dtype).to(self.generated_tokens.device)
        if apply_mask:
            assert step < self.generated_tokens.shape[1]
            cur = self.generated_tokens[:, step, :]
            self.generated_tokens[:, step, :] = torch.where(cur == -1, dec_out, cur)
        else:
            assert step == self.generated_tokens.shape[1]
            self.generated_tokens = torch.cat((self.generated_tokens, dec_out[:, None, :]), dim=1)


def generate_output(config, generated_codes: torch.Tensor, lengths_Bx: torch.Tensor) -> list:
    """reference `_generate_output`, utils.py:301-325: undo the delay pattern, strip the last max_delay frames."""
    md = max(config.codec_delay_pattern)
    cb = revert_audio_delay(generated_codes, config.codec_pad_value, delay_pattern=list(config.codec_delay_pattern))
    cb = cb[:, :-md, :]
    return [cb[i, : int(lengths_Bx[i])].cpu() for i in range(generated_codes.shape[0])]


def preprocess_codec(config, codec) -> torch.Tensor:
    """reference `_preprocess_codec`, mod.py:140-156: prompt codes [T,C] -> delayed [T+max_delay+1,C], BOS prefix,
    one EOS after the data of each channel, PAD afterwards."""
    tok = torch.as_tensor(codec, dtype=torch.long)
    T, C = tok.shape
    delay = list(config.codec_delay_pattern)
    total = T + max(delay) + 1
    t = torch.arange(total).view(total, 1)
    start = torch.tensor(delay).view(1, C) + 1
    rel = t - start                                              # position inside the channel's data
    data = torch.gather(tok, 0, rel.clamp(0, T - 1))
    out = torch.where(rel < 0, torch.full_like(data, config.codec_bos_value), data)
    out = torch.where(rel == T, torch.full_like(data, config.codec_eos_value), out)
    return torch.where(rel > T, torch.full_like(data, config.codec_pad_value), out)
