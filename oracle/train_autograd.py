"""Differentiable CPU restatement of the training step (text model + codec head + loss), torch autograd on CPU.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and bench.py's cpu_baseline leg) -- never by the product package.

Composition of pieces that are each pinned to the reference: oracle/decode.py (RMSNorm, mRoPE, attention: fixtures
attn_*.npz from the transformers classes the reference imports, utils/UniMoE_Audio_model.py:52-56) and
oracle/dcmoe_autograd.py (DCMoE block: fixtures dcmoebwd_*.npz, outputs and gradients of the reference's autograd).
Follows utils/UniMoE_Audio_model.py: decoder layer :210-256, text model :319-457, codec embedding :655-661,
codec head and per-channel shifted cross-entropy + decayed aux weight * mean layer aux :817-854.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import dcmoe_autograd as OA
from . import decode as OD


def forward_loss(cfg, w: Dict[str, torch.Tensor], input_ids, codec_input_ids, attention_mask, codec_labels, aux_weight: float,
                 aux_balance_weight: Optional[torch.Tensor] = None, training: bool = True, forced=None, input_noise=None):
    """-> (loss, codec_loss, aux_mean, last_hidden); differentiable in every tensor of `w` that requires grad.
    input_noise: one [B, T, D] tensor per layer, the samples of the DCMoE input jitter (core.py:243-244; required when training with
    input_jitter_noise > 0: the same samples are injected into the product's blocks)."""
    B, T = input_ids.shape
    D = cfg.hidden_size
    x = w["language_model.embed_tokens.weight"][input_ids]
    if codec_input_ids is not None:
        ce = sum(w[f"codec_embed_tokens.{c}.weight"][codec_input_ids[..., c]] for c in range(cfg.codec_channels))
        m = (input_ids == cfg.codec_placeholder_value).unsqueeze(-1).expand_as(x)
        x = x.masked_scatter(m, ce.to(x.dtype))
    am = torch.ones(B, T, dtype=torch.long) if attention_mask is None else attention_mask.long()
    pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
    key_valid = am.bool()
    hd = cfg.hidden_size // cfg.num_attention_heads
    cos3, sin3 = OD.rope_cos_sin(pos[None].expand(3, -1, -1), hd, cfg.rope_theta, x.dtype)
    cos, sin = OD.mrope_select(cos3, cfg.mrope_section), OD.mrope_select(sin3, cfg.mrope_section)
    pm = None if attention_mask is None else attention_mask.bool()
    abw = None
    if aux_balance_weight is not None:
        abw = attention_mask * aux_balance_weight if attention_mask is not None else aux_balance_weight
    auxes = []
    for l in range(cfg.num_hidden_layers):
        lp = f"language_model.layers.{l}."
        h = OD.rmsnorm(x, w[lp + "input_layernorm.weight"], cfg.rms_norm_eps)
        a, _ = OD.attention(cfg, w, lp + "self_attn.", h, cos, sin, None, key_valid)
        x = x + a
        h = OD.rmsnorm(x, w[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        sub = {k[len(lp + "mlp."):]: v for k, v in w.items() if k.startswith(lp + "mlp.")}
        out = OA.forward(cfg, sub, h, pm, abw, training=training, forced=None if forced is None else forced[l],
                         input_noise=None if input_noise is None else input_noise[l])
        auxes.append(out[5])
        x = x + out[0]
    hs = OD.rmsnorm(x, w["language_model.norm.weight"], cfg.rms_norm_eps)
    C, V = cfg.codec_channels, cfg.codec_vocab_size
    logits = F.linear(hs, w["codec_head.weight"]).float().view(B, T, C, V)
    sl = logits[:, :-1]
    lab = codec_labels[:, 1:]
    codec_loss = None
    for c in range(C):                                       # model.py:830-847: channel 0 always, others when they have labels
        lc = lab[..., c].reshape(-1)
        if c != 0 and not bool((lc != -100).any()):
            continue
        term = F.cross_entropy(sl[:, :, c].reshape(-1, V), lc, ignore_index=-100)
        codec_loss = term if codec_loss is None else codec_loss + term
    aux_mean = torch.stack([a_.float() for a_ in auxes]).mean()
    return codec_loss + aux_weight * aux_mean, codec_loss, aux_mean, hs
