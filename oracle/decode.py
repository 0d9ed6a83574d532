"""CPU restatement of the reference transformer + DAC-token decode loop (torch-CPU).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.

Restates (reference = /root/reference/utils, read as text):
  RMSNorm / mRoPE / GQA attention with KV cache   UniMoE_Audio_model.py:52-56,204-207 -> third-party
      transformers==4.53.1 `Qwen2RMSNorm`, `Qwen2_5_VLRotaryEmbedding`,
      `apply_multimodal_rotary_pos_emb`, `Qwen2_5_VLAttention` (same math in the 5.15.0 copy
      installed here, which pins tests/golden/attn_*.npz)
  decoder layer                                   UniMoE_Audio_model.py:210-256
  text model (layer loop, final norm)             UniMoE_Audio_model.py:319-457
  codec embedding sum / prompt scatter            UniMoE_Audio_model.py:655-670
  _decoder_step (positions, CFG, EOS masks)       UniMoE_Audio_model.py:918-1068
  _sample_next_token                              UniMoE_Audio_model.py:873-916
  generate (EOS countdown, delay padding, packing) UniMoE_Audio_model.py:1070-1231
  delay pattern helpers / DecoderOutput           UniMoE_Audio_utils.py:137-325
  _preprocess_codec                               UniMoE_Audio_mod.py:140-156
The KV cache grows by concatenation per step like the reference's DynamicCache
(UniMoE_Audio_model.py:353-354,1109) so that the timed CPU baseline pays what the reference pays.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .dcmoe import DCMoEOracle

NEG_INF = float("-inf")


# ----------------------------------------------------------------------------- norms / rope
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w * (xf * torch.rsqrt(var + eps)).to(x.dtype)


def rope_cos_sin(position_ids: torch.Tensor, head_dim: int, theta: float, dtype) -> Tuple[torch.Tensor, torch.Tensor]:
    """position_ids [3, rows, T] -> cos, sin [3, rows, T, head_dim] in `dtype`"""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
    freqs = position_ids[..., None].float() * inv_freq            # [3, rows, T, hd/2]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def mrope_select(c: torch.Tensor, section: List[int]) -> torch.Tensor:
    """[3, rows, T, hd] -> [rows, T, hd]: channel block i takes position stream i % 3"""
    parts = c.split(list(section) * 2, dim=-1)
    return torch.cat([p[i % 3] for i, p in enumerate(parts)], dim=-1)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def apply_rope(q, k, cos, sin):
    """q [rows, H, T, hd]; cos/sin [rows, T, hd] (already section-selected)"""
    cos = cos.unsqueeze(1)
    sin = sin.unsqueeze(1)
    return (q * cos) + (rotate_half(q) * sin), (k * cos) + (rotate_half(k) * sin)


# ----------------------------------------------------------------------------- attention
def attention(cfg, w: Dict[str, torch.Tensor], pre: str, x: torch.Tensor, cos, sin, past: Optional[Tuple],
              key_valid: torch.Tensor):
    """x [rows, T, D]; key_valid [rows, L_total] bool (the 2-D attention mask incl. new tokens).
    Returns (out [rows, T, D], (K, V) concatenated caches)."""
    rows, T, D = x.shape
    H, KV, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.hidden_size // cfg.num_attention_heads
    q = F.linear(x, w[pre + "q_proj.weight"], w[pre + "q_proj.bias"]).view(rows, T, H, hd).transpose(1, 2)
    k = F.linear(x, w[pre + "k_proj.weight"], w[pre + "k_proj.bias"]).view(rows, T, KV, hd).transpose(1, 2)
    v = F.linear(x, w[pre + "v_proj.weight"], w[pre + "v_proj.bias"]).view(rows, T, KV, hd).transpose(1, 2)
    q, k = apply_rope(q, k, cos, sin)
    if past is not None:
        k = torch.cat((past[0], k), dim=2)
        v = torch.cat((past[1], v), dim=2)
    L = k.shape[2]
    g = H // KV
    kf = k.float().repeat_interleave(g, dim=1)
    vf = v.float().repeat_interleave(g, dim=1)
    scores = torch.matmul(q.float(), kf.transpose(2, 3)) * (hd ** -0.5)      # [rows, H, T, L]
    qpos = torch.arange(L - T, L).view(1, 1, T, 1)
    kpos = torch.arange(L).view(1, 1, 1, L)
    allowed = (kpos <= qpos) & key_valid.view(rows, 1, 1, L)
    scores = scores.masked_fill(~allowed, NEG_INF)
    p = torch.softmax(scores, dim=-1)
    p = torch.nan_to_num(p, nan=0.0)          # fully masked (left-pad) query rows
    o = torch.matmul(p, vf).to(x.dtype)       # [rows, H, T, hd]
    o = o.transpose(1, 2).reshape(rows, T, H * hd)
    return F.linear(o, w[pre + "o_proj.weight"]), (k, v)


# ----------------------------------------------------------------------------- model
class TextModelOracle:
    """36 x (RMSNorm -> attention -> +res -> RMSNorm -> DCMoE -> +res) -> RMSNorm."""

    def __init__(self, cfg, weights: Dict[str, torch.Tensor], prefix: str = "language_model."):
        self.cfg = cfg
        self.w = weights
        self.p = prefix
        self.moe = [DCMoEOracle(cfg, weights, f"{prefix}layers.{l}.mlp.") for l in range(cfg.num_hidden_layers)]

    def forward(self, x: torch.Tensor, key_valid: torch.Tensor, position_ids: torch.Tensor, cache: Optional[list],
                padding_token_mask: Optional[torch.Tensor] = None, collect_router: bool = False,
                layer_inputs: Optional[list] = None):
        """collect_router: per-layer routing results AND the layer's residual-stream states (x_in, x1 after attention, x_out).
        layer_inputs: teacher forcing for per-layer parity -- layer l reads layer_inputs[l] instead of layer l-1's output (cast to the
        weights' dtype), so an fp32 copy of the model can be walked along the bf16 model's own trajectory one layer at a time."""
        cfg = self.cfg
        hd = cfg.hidden_size // cfg.num_attention_heads
        if position_ids.dim() == 2:
            position_ids = position_ids[None].expand(3, -1, -1)
        cos3, sin3 = rope_cos_sin(position_ids, hd, cfg.rope_theta, x.dtype)
        cos, sin = mrope_select(cos3, cfg.mrope_section), mrope_select(sin3, cfg.mrope_section)
        new_cache, router = [], []
        for l in range(cfg.num_hidden_layers):
            lp = f"{self.p}layers.{l}."
            if layer_inputs is not None:
                x = layer_inputs[l].to(x.dtype)
            x_in = x
            res = x
            h = rmsnorm(x, self.w[lp + "input_layernorm.weight"], cfg.rms_norm_eps)
            a, kv = attention(cfg, self.w, lp + "self_attn.", h, cos, sin, None if cache is None else cache[l], key_valid)
            new_cache.append(kv)
            x = res + a
            res = x
            h = rmsnorm(x, self.w[lp + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
            out = self.moe[l](h, padding_token_mask, None)
            if collect_router:
                router.append(dict(logits=out[1], top_k=out[2], expert_mask=out[3], global_weight=out[4],
                                   aux=out[5], hidden_in=h, hidden_out=out[0], x_in=x_in, x1=res, x_out=res + out[0]))
            x = res + out[0]
        x = rmsnorm(x, self.w[self.p + "norm.weight"], cfg.rms_norm_eps)
        return x, new_cache, router


def codec_embedding(cfg, w, tokens: torch.Tensor) -> torch.Tensor:
    """tokens [..., C] -> sum_c Emb_c[token_c]   (UniMoE_Audio_model.py:655-661)"""
    x = None
    for c in range(cfg.codec_channels):
        e = F.embedding(tokens[..., c].long(), w[f"codec_embed_tokens.{c}.weight"])
        x = e if x is None else x + e
    return x


def input_embedding(cfg, w, input_ids: torch.Tensor, codec_input_ids: Optional[torch.Tensor]) -> torch.Tensor:
    """UniMoE_Audio_model.py:663-670"""
    x = F.embedding(input_ids, w["language_model.embed_tokens.weight"])
    if codec_input_ids is not None:
        ce = codec_embedding(cfg, w, codec_input_ids)
        m = (input_ids == cfg.codec_placeholder_value).unsqueeze(-1).expand_as(x)
        x = x.masked_scatter(m, ce)
    return x


# ----------------------------------------------------------------------------- sampling
def sample_next_token(logits: torch.Tensor, temperature: float, top_p: float, top_k: Optional[int], eos: int,
                      return_probs: bool = False, generator=None):
    """logits [rows, V] fp32 (UniMoE_Audio_model.py:873-916)"""
    if temperature == 0.0:
        return torch.argmax(logits, dim=-1)
    x = logits / temperature
    if eos is not None and eos >= 0:
        top = torch.argmax(x, dim=-1)
        kill = top != eos
        x = x.clone()
        x[kill, eos] = NEG_INF
    if top_k is not None:
        _, idx = torch.topk(x, k=top_k, dim=-1)
        keep = torch.zeros_like(x, dtype=torch.bool).scatter(-1, idx, True)
        x = x.masked_fill(~keep, NEG_INF)
    if top_p < 1.0:
        pr = torch.softmax(x, dim=-1)
        sp, si = torch.sort(pr, dim=-1, descending=True)
        cum = torch.cumsum(sp, dim=-1)
        rm = cum > top_p
        rm = torch.roll(rm, 1, -1)
        rm[..., 0] = False
        x = x.masked_fill(torch.zeros_like(rm).scatter(-1, si, rm), NEG_INF)
    probs = torch.softmax(x, dim=-1)
    if return_probs:
        return probs
    return torch.multinomial(probs, 1, generator=generator).squeeze(-1)


def cfg_and_mask(cfg, logits_2B: torch.Tensor, cfg_scale: float, enable_eos: bool, eos_mul: float) -> torch.Tensor:
    """logits_2B [2B, C, V] fp32 -> guided/masked [B, C, V]  (UniMoE_Audio_model.py:991-1017)"""
    B = logits_2B.shape[0] // 2
    pair = logits_2B.view(B, 2, *logits_2B.shape[1:])
    if cfg_scale != 0:
        un, co = pair[:, 0], pair[:, 1]
        out = co + cfg_scale * (co - un)
    else:
        out = pair[:, 1].clone()
    eos = cfg.codec_eos_value
    if enable_eos:
        out[:, :, eos + 1:] = NEG_INF
        out[:, 1:, eos:] = NEG_INF
    else:
        out[:, :, eos:] = NEG_INF
    out[:, 0, eos] *= eos_mul
    return out


# ----------------------------------------------------------------------------- delay pattern
def apply_delay(codes: torch.Tensor, delay: List[int], pad: int, bos: int) -> torch.Tensor:
    """codes [B, T, C]; out[b,t,c] = codes[b, t-delay_c, c], BOS where t-delay_c < 0, PAD where >= T
    (UniMoE_Audio_utils.py:137-183)"""
    B, T, C = codes.shape
    t = torch.arange(T).view(1, T, 1) - torch.tensor(delay).view(1, 1, C)
    g = torch.gather(codes, 1, t.clamp(0, T - 1).expand(B, T, C))
    return torch.where(t < 0, torch.tensor(bos, dtype=codes.dtype),
                       torch.where(t >= T, torch.tensor(pad, dtype=codes.dtype), g))


def revert_delay(codes: torch.Tensor, delay: List[int], pad: int) -> torch.Tensor:
    """out[b,t,c] = codes[b, min(t+delay_c, T-1), c]; PAD where the (clamped) index >= T: never, kept for parity
    (UniMoE_Audio_utils.py:186-227)"""
    B, T, C = codes.shape
    t = torch.minimum(torch.arange(T).view(1, T, 1) + torch.tensor(delay).view(1, 1, C), torch.tensor(T - 1))
    g = torch.gather(codes, 1, t.expand(B, T, C))
    return torch.where(t >= T, torch.tensor(pad, dtype=codes.dtype), g)


def prepare_audio_prompt(cfg, audio_prompts: list) -> Tuple[torch.Tensor, List[int]]:
    """UniMoE_Audio_utils.py:230-268"""
    C, bos, delay = cfg.codec_channels, cfg.codec_bos_value, cfg.codec_delay_pattern
    B = len(audio_prompts)
    T = max((p.shape[0] if p is not None else 0) for p in audio_prompts) + max(delay) + 1
    buf = torch.full((B, T, C), -1, dtype=torch.int32)
    buf[:, 0, :] = bos
    steps = []
    for i, p in enumerate(audio_prompts):
        if p is not None:
            buf[i, 1: p.shape[0] + 1] = p.to(torch.int32)
            steps.append(p.shape[0] + 1)
        else:
            steps.append(1)
    return apply_delay(buf, delay, -1, bos), steps


def generate_output(cfg, codes: torch.Tensor, lengths: torch.Tensor) -> list:
    """UniMoE_Audio_utils.py:301-325"""
    md = max(cfg.codec_delay_pattern)
    cb = revert_delay(codes, cfg.codec_delay_pattern, cfg.codec_pad_value)[:, :-md, :]
    return [cb[i, : int(lengths[i])] for i in range(codes.shape[0])]


def preprocess_codec(cfg, codec) -> torch.Tensor:
    """prompt codes [T,12] -> delayed [T+md+1,12] with BOS prefix, EOS then PAD (UniMoE_Audio_mod.py:140-156)"""
    tok = torch.as_tensor(codec, dtype=torch.long)
    T, md = tok.shape[0], max(cfg.codec_delay_pattern)
    out = torch.zeros((T + md + 1, cfg.codec_channels), dtype=torch.long)
    for c, d in enumerate(cfg.codec_delay_pattern):
        st = d + 1
        out[:st, c] = cfg.codec_bos_value
        out[st: st + T, c] = tok[:, c]
        out[st + T:, c] = cfg.codec_pad_value
        if st + T < out.shape[0]:
            out[st + T, c] = cfg.codec_eos_value
    return out


# ----------------------------------------------------------------------------- generate
class GenerateOracle:
    """Restates generate()/_decoder_step() with an injectable language model:
    lm(inputs_embeds [rows,T,D], key_valid [rows,L], position_ids [rows,T], cache) -> (hidden, cache)."""

    def __init__(self, cfg, weights, lm=None):
        self.cfg = cfg
        self.w = weights
        if lm is None:
            tm = TextModelOracle(cfg, weights)

            def lm(x, key_valid, pos, cache):
                h, c, _ = tm.forward(x, key_valid, pos, cache)
                return h, c
        self.lm = lm
        self.trace = []

    def generate(self, input_ids, attention_mask, prefill_tokens, prefill_steps, max_tokens, min_tokens=None,
                 codec_input_ids=None, cfg_scale=3.0, temperature=1.2, top_p=0.95, cfg_filter_top_k=45,
                 eos_prob_mul_factor=0.8, do_sample=True, generator=None):
        cfg, w = self.cfg, self.w
        B = input_ids.shape[0] // 2
        eos, pad = cfg.codec_eos_value, cfg.codec_pad_value
        delay = torch.tensor(cfg.codec_delay_pattern, dtype=torch.long)
        md = int(delay.max())
        tokens = prefill_tokens.clone()                 # [B, T0, C] int32, -1 = to be generated
        dec_step = min(prefill_steps) - 1
        eos_detected = torch.zeros(B, dtype=torch.bool)
        countdown = torch.full((B,), -1, dtype=torch.long)
        finished = torch.full((B,), -1, dtype=torch.long)
        bos_over = False
        key_valid = attention_mask.bool()
        pos = attention_mask.long().cumsum(-1) - 1
        pos = pos.masked_fill(attention_mask == 0, 1)
        x = input_embedding(cfg, w, input_ids, codec_input_ids)
        _, cache = self.lm(x, key_valid, pos, None)     # prefill (:1116-1133)
        key_valid = torch.cat([key_valid, torch.ones((2 * B, 1), dtype=torch.bool)], dim=-1)
        while dec_step < max_tokens:
            if bool((countdown == 0).all()):
                break
            cur = dec_step + 1
            tok = tokens[:, dec_step: dec_step + 1, :]                         # [B,1,C]
            pos = (key_valid.long().cumsum(-1) - 1).masked_fill(~key_valid, 1)[:, -1:]
            tok2 = tok.repeat_interleave(2, dim=0)                              # CFG rows (:945)
            h, cache = self.lm(codec_embedding(cfg, w, tok2), key_valid, pos, cache)
            logits = F.linear(h, w["codec_head.weight"]).float()               # (:982)
            logits = logits.view(2 * B, -1, cfg.codec_channels, cfg.codec_vocab_size)[:, -1]
            key_valid = torch.cat([key_valid, torch.ones((2 * B, 1), dtype=torch.bool)], dim=-1)
            enable_eos = (min_tokens is None) or (dec_step >= min_tokens)
            guided = cfg_and_mask(cfg, logits, cfg_scale, enable_eos, eos_prob_mul_factor)
            flat = guided.reshape(B * cfg.codec_channels, -1)
            if do_sample:
                pred = sample_next_token(flat, temperature, top_p, cfg_filter_top_k, eos, generator=generator)
            else:
                pred = torch.argmax(flat, dim=1)
            pred = pred.view(B, cfg.codec_channels).clone()
            # EOS detection / countdown (:1173-1197)
            active = countdown != 0
            trig = torch.zeros(B, dtype=torch.bool)
            is_max = cur >= max_tokens - md
            trig[active] = ((~eos_detected[active]) & (pred[active, 0] == eos)) | is_max
            eos_detected |= trig
            start = trig & (countdown < 0)
            countdown[start] = md
            finished[start] = cur
            padding = countdown > 0
            if bool(padding.any()):
                after = (md - countdown[padding]).unsqueeze(1)
                pa = pred[padding]
                pa[after == delay.unsqueeze(0)] = eos
                pa[after > delay.unsqueeze(0)] = pad
                pred[padding] = pa
                countdown[padding] -= 1
            if not bos_over:
                bos_over = all(cur - ps >= md for ps in prefill_steps)
            # DecoderOutput.update_one (UniMoE_Audio_utils.py:290-298)
            pi = pred.to(tokens.dtype)
            if not bos_over:
                assert cur < tokens.shape[1]
                keep = tokens[:, cur] != -1
                tokens[:, cur] = torch.where(keep, tokens[:, cur], pi)
            else:
                assert cur == tokens.shape[1]
                tokens = torch.cat((tokens, pi[:, None]), dim=1)
            self.trace.append(pred.clone())
            dec_step += 1
        final = dec_step + 1
        finished[finished == -1] = final - md
        lengths = torch.clamp(finished - torch.tensor(prefill_steps), min=0)
        max_len = int(lengths.max()) + md
        if max_len <= 0:
            return None, None
        out = torch.full((B, max_len, cfg.codec_channels), pad, dtype=torch.long)
        for i in range(B):
            n = int(lengths[i]) + md
            if n > 0:
                seg = tokens[i, prefill_steps[i]: prefill_steps[i] + n]
                out[i, : seg.shape[0]] = seg
        self.tokens = tokens
        return out, lengths
