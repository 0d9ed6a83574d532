"""Model-level mirror of the reference API on the HIP engine.

`UniAudioRVQQwen2_5VLMoEForConditionalGeneration` keeps the reference's parameter names
(reference utils/UniMoE_Audio_model.py:296-311,460-487) so reference checkpoints load by name, and its
`generate()` keeps the reference signature (model.py:1070-1091) and return value (`codes[B,T,C] int64, lengths[B]`).
Underneath, prefill and every decode step run in libumoe_hip.so (`umoe_engine_*`): one host call per step,
replayed as a hipGraph, no per-step host synchronisation (the reference has > 500).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .codec_utils import DecoderOutput
from .config import UniMoEAudioConfig
from .dcmoe import UniMoEAudioSparseMoeBlock


@dataclass
class BaseModelOutputWithPast:
    """reference model.py:180-190 (same field names)"""
    last_hidden_state: torch.Tensor = None
    past_key_values: Optional[tuple] = None
    hidden_states: Optional[tuple] = None
    attentions: Optional[tuple] = None
    all_router_logits: Optional[tuple] = None
    all_router_top_k: Optional[tuple] = None
    all_router_weight: Optional[tuple] = None
    all_router_expert_mask: Optional[tuple] = None
    all_aux_loss: Optional[tuple] = None


@dataclass
class MoEQwen2_5VLCausalLMOutputWithPast:
    """reference model.py:165-177 (+ codec_logits, which the reference only uses internally for the loss)"""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    codec_logits: Optional[torch.Tensor] = None
    past_key_values: Optional[tuple] = None
    hidden_states: Optional[torch.Tensor] = None
    attentions: Optional[tuple] = None
    rope_deltas: Optional[torch.Tensor] = None
    all_router_logits: Optional[tuple] = None
    all_router_top_k: Optional[tuple] = None
    all_router_expert_mask: Optional[tuple] = None
    all_router_weight: Optional[tuple] = None
    aux_balance_loss: Optional[torch.Tensor] = None


class Qwen2RMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps


class Qwen2_5_VLAttention(nn.Module):
    """parameter container with the transformers names (q/k/v with bias, o without)."""

    def __init__(self, config):
        super().__init__()
        D, H, KV, hd = config.hidden_size, config.num_attention_heads, config.num_key_value_heads, config.head_dim
        self.q_proj = nn.Linear(D, H * hd, bias=True)
        self.k_proj = nn.Linear(D, KV * hd, bias=True)
        self.v_proj = nn.Linear(D, KV * hd, bias=True)
        self.o_proj = nn.Linear(H * hd, D, bias=False)


class Qwen2_5_VLMoEDecoderLayer(nn.Module):
    """reference model.py:193-208"""

    def __init__(self, config, layer_idx: int):
        super().__init__()
        self.self_attn = Qwen2_5_VLAttention(config)
        self.mlp = UniMoEAudioSparseMoeBlock(config)
        self.input_layernorm = Qwen2RMSNorm(config.hidden_size, config.rms_norm_eps)
        self.post_attention_layernorm = Qwen2RMSNorm(config.hidden_size, config.rms_norm_eps)


class Qwen2_5_VLMoETextModel(nn.Module):
    """reference model.py:296-311"""

    def __init__(self, config):
        super().__init__()
        self.embed_tokens = nn.Embedding(config.vocab_size, config.hidden_size)
        self.layers = nn.ModuleList([Qwen2_5_VLMoEDecoderLayer(config, i) for i in range(config.num_hidden_layers)])
        self.norm = Qwen2RMSNorm(config.hidden_size, config.rms_norm_eps)


class UniAudioRVQQwen2_5VLMoEForConditionalGeneration(nn.Module):
    def __init__(self, config: UniMoEAudioConfig, with_lm_head: bool = False):
        super().__init__()
        self.config = config
        self.language_model = Qwen2_5_VLMoETextModel(config)
        self.num_channels = config.codec_channels
        self.codec_vocab_size = config.codec_vocab_size
        self.codec_embed_tokens = nn.ModuleList(
            [nn.Embedding(self.codec_vocab_size, config.hidden_size) for _ in range(self.num_channels)])
        self.codec_placeholder_value = config.codec_placeholder_value
        self.codec_head = nn.Linear(config.hidden_size, self.num_channels * self.codec_vocab_size, bias=False)
        if with_lm_head:  # unused on the audio-token path (computed and discarded by the reference, model.py:817)
            self.lm_head = nn.Linear(config.hidden_size, config.vocab_size, bias=False)
        if getattr(config, "vision_config", None):      # reference model.py:476-478: `self.visual`, same parameter names
            from .vision import Qwen2_5_VisionTransformerPretrainedModel
            self.visual = Qwen2_5_VisionTransformerPretrainedModel(config.vision_config)
        self._engine: Optional["DecodeEngine"] = None

    @property
    def device(self):
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, local_dir: str, torch_dtype=torch.bfloat16, attn_implementation: Optional[str] = None, device=None,
                        ep_rank: int = 0, ep_size: int = 1, config=None):
        """Reference signature (utils/UniMoE_Audio_mod.py:79-92) for local checkpoint directories; see checkpoint.py."""
        from . import checkpoint
        return checkpoint.from_pretrained(cls, local_dir, torch_dtype=torch_dtype, attn_implementation=attn_implementation,
                                          device=device, ep_rank=ep_rank, ep_size=ep_size, config=config)

    @torch.no_grad()
    def init_synthetic(self, seed: int = 1234, std: Optional[float] = None):
        """N(0, initializer_range^2) Linear/Embedding weights, unit RMSNorm, zero biases; seed + layer index per
        layer (BASELINE.md measurement plan).  Works on whatever device the parameters live on."""
        std = self.config.initializer_range if std is None else std
        dev = self.device
        g = torch.Generator(device=dev)

        def fill(mod, s):
            g.manual_seed(s)
            for n, p in mod.named_parameters():
                if n.endswith("layernorm.weight"):
                    p.fill_(1.0)
                elif n.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(0, std, generator=g)
        for i, layer in enumerate(self.language_model.layers):
            fill(layer, seed + i)
        g.manual_seed(seed + 1000)
        self.language_model.embed_tokens.weight.normal_(0, std, generator=g)
        self.language_model.norm.weight.fill_(1.0)
        for e in self.codec_embed_tokens:
            e.weight.normal_(0, std, generator=g)
        self.codec_head.weight.normal_(0, std, generator=g)
        return self

    # ---- input embeddings (reference model.py:655-670): gathers + one scatter, device plumbing -------------
    def codec_embedding(self, codec_input_ids: torch.Tensor) -> torch.Tensor:
        emb = torch.stack([e.weight for e in self.codec_embed_tokens], 0)
        flat = codec_input_ids.reshape(-1, self.num_channels)
        return ops.codec_embed_sum(flat, emb.contiguous()).reshape(*codec_input_ids.shape[:-1], -1)

    def calculate_input_embedding(self, input_ids, codec_input_ids):
        x = self.language_model.embed_tokens.weight[input_ids]
        if codec_input_ids is not None:
            ce = self.codec_embedding(codec_input_ids)
            m = (input_ids == self.codec_placeholder_value).unsqueeze(-1).expand_as(x)
            x = x.masked_scatter(m, ce)
        return x

    # ---- multimodal prompt pieces (reference model.py:513-652,634-652,708-751) -------------------------------------
    def _vision(self):
        v = getattr(self, "visual", None)
        if v is None:
            raise NotImplementedError("this model was built without a vision tower (config.vision_config is None)")
        return v

    def get_video_features(self, pixel_values_videos: torch.Tensor, video_grid_thw: torch.Tensor):
        v = self._vision()
        emb = v(pixel_values_videos.to(self.device, v.dtype), grid_thw=video_grid_thw)
        return torch.split(emb, (video_grid_thw.prod(-1) // v.spatial_merge_size ** 2).tolist())

    def get_image_features(self, pixel_values: torch.Tensor, image_grid_thw: torch.Tensor):
        v = self._vision()
        emb = v(pixel_values.to(self.device, v.dtype), grid_thw=image_grid_thw)
        return torch.split(emb, (image_grid_thw.prod(-1) // v.spatial_merge_size ** 2).tolist())

    def get_rope_index(self, input_ids=None, image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None, attention_mask=None):
        from .vision import rope_index
        cfg = self.config
        vc = cfg.vision_config or {}
        return rope_index(input_ids, image_grid_thw, video_grid_thw, second_per_grid_ts, attention_mask,
                          spatial_merge_size=vc.get("spatial_merge_size", 2), tokens_per_second=vc.get("tokens_per_second", 2),
                          image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_token_id)

    def multimodal_embedding(self, input_ids, codec_input_ids=None, pixel_values=None, image_grid_thw=None, pixel_values_videos=None,
                             video_grid_thw=None):
        """text / codec embeddings with the image and video embeddings scattered over their placeholder tokens (the working path of
        the reference's forward, model.py:706-751; its generate() drops the pixels in **kwargs)"""
        from .vision import scatter_vision_embeddings
        x = self.calculate_input_embedding(input_ids, codec_input_ids)
        if pixel_values is not None:
            x = scatter_vision_embeddings(x, input_ids, self.config.image_token_id, torch.cat(self.get_image_features(pixel_values, image_grid_thw), 0), "Image")
        if pixel_values_videos is not None:
            x = scatter_vision_embeddings(x, input_ids, self.config.video_token_id, torch.cat(self.get_video_features(pixel_values_videos, video_grid_thw), 0), "Video")
        return x

    # ---- packed weights shared by forward() and the decode engine ------------------------------------------
    def packed(self) -> dict:
        key = tuple((p.data_ptr(), p._version) for n, p in self.named_parameters() if not n.startswith("visual."))
        if getattr(self, "_pk", None) is not None and self._pk_key == key:
            return self._pk
        layers = []
        for layer in self.language_model.layers:
            a = layer.self_attn
            qkv_rm = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).contiguous()
            layers.append(dict(
                qkv_rm=qkv_rm, qkv_w=ops.pack_weight(qkv_rm),
                qkv_b=torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0).float().contiguous(),
                o_w=ops.pack_weight(a.o_proj.weight.data.contiguous()), moe=layer.mlp.prepare()))
        self._pk = dict(layers=layers, head=ops.pack_weight(self.codec_head.weight.data.contiguous()),
                        emb=torch.stack([e.weight.data for e in self.codec_embed_tokens], 0).contiguous())
        self._pk_key = key
        return self._pk

    # ---- full-sequence forward (no KV cache kept): reference Qwen2_5_VLMoETextModel.forward, model.py:319-457 ------
    @torch.no_grad()
    def text_forward(self, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                     position_ids: Optional[torch.Tensor] = None, padding_token_mask: Optional[torch.Tensor] = None,
                     aux_balance_weight: Optional[torch.Tensor] = None, output_router_logits_and_topk: bool = False, kv_sink=None):
        """kv_sink(layer, k [B, KVH, T, hd], v): called with every layer's roped keys / values (the buffers are reused by the next layer:
        copy what you keep) -- how an expert-parallel engine that holds only its local experts gets its KV cache from this forward."""
        cfg, dev = self.config, inputs_embeds.device
        B, T, D = inputs_embeds.shape
        H, KVH, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
        am = torch.ones(B, T, dtype=torch.long, device=dev) if attention_mask is None else attention_mask.to(dev).long()
        if position_ids is None:                       # text-only mRoPE: three equal streams (model.py:365-368)
            pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
            position_ids = pos[None].expand(3, -1, -1)
        elif position_ids.dim() == 2:
            position_ids = position_ids[None].expand(3, -1, -1)
        pos3 = position_ids.reshape(3, B * T).to(torch.int32).contiguous()
        kv_pos = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
        first_valid = (am != 0).float().argmax(-1).to(torch.int32).contiguous()   # keys before it are left padding
        q0 = torch.zeros(B, dtype=torch.int32, device=dev)
        cos, sin = ops.rope_tables(int(position_ids.max()) + 2, hd, cfg.rope_theta, dev)
        pk = self.packed()
        x = inputs_embeds.reshape(B * T, D).contiguous()
        layers = self.language_model.layers
        h = ops.rmsnorm(x, layers[0].input_layernorm.weight.data, cfg.rms_norm_eps)
        kc = torch.empty((B, KVH, T, hd), dtype=torch.bfloat16, device=dev)
        vc = torch.empty_like(kc)
        stats = dict(logits=[], top_k=[], expert_mask=[], weight=[], aux=[])
        for li, layer in enumerate(layers):
            lp = pk["layers"][li]
            tiled = B * T >= 64                      # compute-bound: tiled MFMA kernel on the row-major tensors
            qkv = (ops.tlinear(h, lp["qkv_rm"], bias=lp["qkv_b"]) if tiled
                   else ops.linear(h, lp["qkv_w"], (H + 2 * KVH) * hd, bias=lp["qkv_b"]))
            q = ops.qkv_mrope_kvappend(qkv, cos, sin, pos3, kv_pos, T, H, KVH, hd, cfg.mrope_section, kc, vc)
            if kv_sink is not None:
                kv_sink(li, kc, vc)
            ao = ops.attention(q, kc, vc, first_valid, q0, T, H, splits=1)
            x1 = (ops.tlinear(ao, layer.self_attn.o_proj.weight.data, resid=x) if tiled
                  else ops.linear(ao, lp["o_w"], D, resid=x))                             # model.py:238
            h2 = ops.rmsnorm(x1, layer.post_attention_layernorm.weight.data, cfg.rms_norm_eps)
            out = layer.mlp(h2.view(B, T, D), padding_token_mask, aux_balance_weight)    # model.py:241
            nxt = layers[li + 1].input_layernorm.weight.data if li + 1 < len(layers) else self.language_model.norm.weight.data
            h, x = ops.rmsnorm(out[0].reshape(B * T, D), nxt, cfg.rms_norm_eps, resid=x1)  # x = x1 + moe (model.py:242)
            if output_router_logits_and_topk:
                stats["logits"].append(out[1]); stats["top_k"].append(out[2])
            stats["expert_mask"].append(out[3]); stats["weight"].append(out[4]); stats["aux"].append(out[5])
        return BaseModelOutputWithPast(last_hidden_state=h.view(B, T, D), all_router_logits=tuple(stats["logits"]) or None,
                                       all_router_top_k=tuple(stats["top_k"]) or None,
                                       all_router_expert_mask=tuple(stats["expert_mask"]), all_router_weight=tuple(stats["weight"]),
                                       all_aux_loss=tuple(stats["aux"]))

    @property
    def cur_aux_weight(self):
        """linear decay l_aux_weight -> min_l_aux_weight over l_aux_weight_decay_steps (model.py:489-493)"""
        cfg = self.config
        steps = max(1, cfg.l_aux_weight_decay_steps)
        ts = getattr(self, "training_steps", 0)
        if ts >= steps:
            return cfg.min_l_aux_weight
        return cfg.l_aux_weight - (cfg.l_aux_weight - cfg.min_l_aux_weight) / steps * ts

    @torch.no_grad()
    def forward(self, input_ids=None, codec_input_ids=None, attention_mask=None, position_ids=None, inputs_embeds=None,
                labels=None, codec_labels=None, aux_balance_weight=None, padding_token_mask=None,
                output_router_logits_and_topk=None, **unused):
        """reference UniAudioRVQQwen2_5VLMoEForConditionalGeneration.forward (model.py:672-871), forward pass only:
        embeddings -> text model -> codec head -> 12 shifted per-channel CE terms + decayed aux weight * mean layer aux."""
        dev = self.device
        if inputs_embeds is None:
            inputs_embeds = self.calculate_input_embedding(input_ids.to(dev), None if codec_input_ids is None else codec_input_ids.to(dev))
        if attention_mask is not None:
            attention_mask = attention_mask.to(dev)
            if aux_balance_weight is not None:
                aux_balance_weight = attention_mask * aux_balance_weight.to(dev)          # model.py:793-794
            if padding_token_mask is None:
                padding_token_mask = attention_mask.bool()                                 # model.py:796-797
        out = self.text_forward(inputs_embeds, attention_mask, position_ids, padding_token_mask, aux_balance_weight,
                                bool(output_router_logits_and_topk))
        B, T, D = out.last_hidden_state.shape
        C, V = self.num_channels, self.codec_vocab_size
        hs = out.last_hidden_state.reshape(B * T, D).contiguous()
        codec_logits = (ops.tlinear(hs, self.codec_head.weight.data, out_f32=True) if B * T >= 64
                        else ops.linear(hs, self.packed()["head"], C * V, out_f32=True)).view(B, T, C, V)   # model.py:818-819
        loss = aux_mean = None
        if labels is not None and codec_labels is not None:
            aux_mean = torch.stack([a.float() for a in out.all_aux_loss]).mean()            # model.py:824-826
            sl = codec_logits[:, :-1].reshape(B * (T - 1), C, V).contiguous()
            lab = codec_labels.to(dev)[:, 1:].reshape(B * (T - 1), C).contiguous()
            codec_loss, _, _ = ops.codec_ce(sl, lab)
            loss = codec_loss + self.cur_aux_weight * aux_mean
            self.training_steps = getattr(self, "training_steps", 0) + 1                    # model.py:827
        return MoEQwen2_5VLCausalLMOutputWithPast(loss=loss, logits=None, codec_logits=codec_logits,
                                                  hidden_states=out.last_hidden_state, all_router_logits=out.all_router_logits,
                                                  all_router_top_k=out.all_router_top_k,
                                                  all_router_expert_mask=out.all_router_expert_mask,
                                                  all_router_weight=out.all_router_weight, aux_balance_loss=aux_mean)

    def _run_with_labels(self, eng, dec_output, max_tokens, min_tokens, cfg_scale, eos_mul, debug_guidance_step, use_graph):
        """generate() with teacher labels in the DecoderOutput (reference model.py:1138-1143,1155,1019-1048,1170-1171): every step
        prints the "golden loss" of the labels under the guided logits, and the first `debug_guidance_step` steps (all of them
        for -1) feed the LABELS forward instead of the sample.  A diagnostic path: one host round trip per step."""
        cfg = self.config
        B, C, V, eos = eng.batch, cfg.codec_channels, cfg.codec_vocab_size, cfg.codec_eos_value
        labels = dec_output.labels_prefill.to(self.device)
        step0 = int(eng.state[4 * B].item())
        self.golden_losses = []
        budget = max_tokens - step0
        for i in range(budget):
            dec_step = step0 + i
            lab = labels[:, dec_step + 1] if dec_step + 1 < labels.shape[1] else None
            guided_step = lab is not None and (dec_step < debug_guidance_step or debug_guidance_step == -1)
            if guided_step:                          # pred = labels: the masked update keeps what is already in the token buffer
                cur = eng.tokens[:, dec_step + 1]
                eng.tokens[:, dec_step + 1] = torch.where(cur == -1, lab.to(torch.int32), cur)
            eng.step(use_graph)
            if lab is not None:
                lg = eng.copy_buffer("logits", torch.float32, (2 * B, C, V)).view(B, 2, C, V)
                guided = (lg[:, 1] + cfg_scale * (lg[:, 1] - lg[:, 0])) if cfg_scale != 0 else lg[:, 1].clone()      # model.py:994-999
                if min_tokens is None or dec_step >= min_tokens:                                                     # enable_eos
                    guided[:, :, eos + 1:] = float("-inf")
                    guided[:, 1:, eos:] = float("-inf")
                else:
                    guided[:, :, eos:] = float("-inf")
                guided[:, 0, eos] *= eos_mul
                gl = golden_loss(guided, lab, eos)
                self.golden_losses.append(None if gl is None else float(gl))
                print(f"golden loss: {gl}")                                                                          # model.py:1048
            if eng.all_done():
                break

    # ---- engine -----------------------------------------------------------------------------------------
    def _pack_key(self):
        return tuple((p.data_ptr(), p._version) for n, p in self.named_parameters() if not n.startswith("visual."))

    def engine(self, batch: int, max_prompt: int, max_tokens: int, attn_splits: int = 8, ep=None) -> "DecodeEngine":
        """The decode engine for this shape, rebuilt when the shape, the weights (data pointer / version of any parameter) or the
        expert-parallel link changed.  `ep`: an unimoe_audio_amd.ep.EpLink (every rank of the link calls this together)."""
        need_L = max_prompt + max_tokens + 8
        e = self._engine
        key = self._pack_key()
        if e is None or e.batch != batch or e.Lmax < need_L or e.Tmax < max_tokens + 64 or e.pack_key != key or e.ep is not ep:
            if e is not None:
                e.close()
            self._engine = DecodeEngine(self, batch, Lmax=need_L, Tmax=max_tokens + 64, attn_splits=attn_splits, ep=ep)
        return self._engine

    @torch.no_grad()
    def generate(self, input_ids, attention_mask, dec_output: DecoderOutput, max_tokens, min_tokens=None,
                 codec_input_ids: Optional[torch.Tensor] = None, pixel_values=None, pixel_values_videos=None,
                 image_grid_thw=None, video_grid_thw=None, second_per_grid_ts=None, cfg_scale: float = 3.0,
                 temperature: float = 1.2, top_p: float = 0.95, cfg_filter_top_k: int = 45,
                 eos_prob_mul_factor: float = 0.8, do_sample: bool = True, debug_guidance_step: int = 0, use_cache=True,
                 seed: int = 0, use_graph: bool = True, poll_every: int = 16, vision_in_generate: bool = False):
        """reference generate(), utils/UniMoE_Audio_model.py:1070-1231 (same arguments, same return).
        vision_in_generate (not in the reference).  The reference's generate() accepts pixel_values(_videos) but never uses them: it builds
        inputs_embeds with calculate_input_embedding only (model.py:1116: the <|video_pad|> / <|image_pad|> tokens keep their TEXT
        embeddings), passes explicit 1-D positions cumsum(attention_mask) - 1 at the prefill (:1113-1114) and at every decode step
        (:940-944), and its text model drops the pixel arguments.  False (default) reproduces exactly that, so video_text_to_music
        yields the tokens the reference's inference path yields; True runs what the reference's forward() does for training
        (model.py:708-790): vision tower, embeddings scattered over the pad tokens, 3-D mRoPE positions from get_rope_index.
        use_cache=False: the reference recomputes the whole prefix every step without a cache (model.py:964-980) and allows it only
        without a codec prompt (:1092-1093); the result is the same tokens, so the engine serves it from its KV cache."""
        if not use_cache and codec_input_ids is not None:
            raise AssertionError("use_cache=False with a codec prompt: the reference asserts use_cache here (model.py:1092-1093)")
        dev = self.device
        input_ids, attention_mask = input_ids.to(dev), attention_mask.to(dev)
        B = input_ids.shape[0] // 2
        T = input_ids.shape[1]
        eng = self.engine(B, T, int(max_tokens))
        pos3 = deltas = None
        if vision_in_generate and (pixel_values is not None or pixel_values_videos is not None):
            x = self.multimodal_embedding(input_ids, None if codec_input_ids is None else codec_input_ids.to(dev), pixel_values, image_grid_thw,
                                          pixel_values_videos, video_grid_thw)
            pos3, deltas = self.get_rope_index(input_ids, image_grid_thw, video_grid_thw, second_per_grid_ts, attention_mask)   # model.py:753-777
        else:
            x = self.calculate_input_embedding(input_ids, None if codec_input_ids is None else codec_input_ids.to(dev))
        eng.prefill(x.reshape(-1, x.shape[-1]).contiguous(), attention_mask, position_ids=pos3, rope_deltas=deltas)
        eng.start_decode(dec_output.generated_tokens, dec_output.prefill_steps, int(max_tokens), min_tokens,
                         cfg_scale=cfg_scale, temperature=temperature, top_p=top_p, top_k=cfg_filter_top_k,
                         eos_mul=eos_prob_mul_factor, do_sample=do_sample, seed=seed)
        if getattr(dec_output, "labels_prefill", None) is not None:
            self._run_with_labels(eng, dec_output, int(max_tokens), min_tokens, cfg_scale, eos_prob_mul_factor, int(debug_guidance_step),
                                  use_graph)
        else:
            eng.run(use_graph=use_graph, poll_every=poll_every)
        codes, lengths, tokens = eng.finish()
        dec_output.generated_tokens = tokens
        if codes is None:
            print("Warning: Nothing generated for any sequence in the batch.")        # model.py:1230
        return codes, lengths


def golden_loss(guided_BxCxV: torch.Tensor, labels_BxC: torch.Tensor, eos: int) -> Optional[torch.Tensor]:
    """The teacher's loss of one decode step (reference _decoder_step, model.py:1019-1048): labels above EOS are ignored on channel 0,
    labels >= EOS on the delayed channels; CrossEntropy per channel on the GUIDED logits (after CFG, masks and the EOS factor),
    channel 0 weighted 3, channels without a valid label skipped (except channel 0)."""
    lab = labels_BxC.clone().long()
    lab[lab > eos] = -100
    rest = lab[:, 1:]
    rest[rest >= eos] = -100
    total = None
    for c in range(lab.shape[1]):
        if c != 0 and int((lab[:, c] != -100).sum()) == 0:
            continue
        l = torch.nn.functional.cross_entropy(guided_BxCxV[:, c].float(), lab[:, c], ignore_index=-100) * (3 if c == 0 else 1)
        total = l if total is None else total + l
    return total


class DecodeEngine:
    """Python face of umoe_engine_*: packs the weights once, owns the C engine and the decode state."""

    def __init__(self, model: UniAudioRVQQwen2_5VLMoEForConditionalGeneration, batch: int, Lmax: int, Tmax: int,
                 attn_splits: int = 8, max_pos: Optional[int] = None, ep=None, ep_connect: bool = True):
        cfg = model.config
        dev = model.device
        if dev.type != "cuda":
            raise L.UmoeError("DecodeEngine needs the model on a ROCm device; there is no CPU path in the product")
        self.model, self.cfg, self.dev = model, cfg, dev
        self.batch, self.rows, self.Lmax, self.Tmax = batch, 2 * batch, int(Lmax), int(Tmax)
        sec = list(cfg.mrope_section)
        c = L.EngineCfg(hidden=cfg.hidden_size, layers=cfg.num_hidden_layers, heads=cfg.num_attention_heads,
                        kv_heads=cfg.num_key_value_heads, head_dim=cfg.head_dim, n_dyn=cfg.num_dyn,
                        n_real=cfg.mlp_dynamic_expert_num, n_fix=cfg.mlp_fixed_expert_num,
                        inter_dyn=cfg.dynamic_intermediate_size, inter_shared=cfg.shared_intermediate_size,
                        codec_channels=cfg.codec_channels, codec_vocab=cfg.codec_vocab_size, eos=cfg.codec_eos_value,
                        pad=cfg.codec_pad_value, bos=cfg.codec_bos_value, mrope0=sec[0], mrope1=sec[1], mrope2=sec[2],
                        rms_eps=cfg.rms_norm_eps, top_p=float(cfg.mlp_dynamic_top_p), fixed_top_k=int(cfg.mlp_dynamic_top_k),
                        jitter_eps=float(cfg.router_jitter_noise), rows=self.rows, Lmax=self.Lmax, Tmax=self.Tmax,
                        attn_splits=attn_splits, ep_rank=0 if ep is None else ep.rank, ep_size=1 if ep is None else ep.size)
        self.ep = ep
        self.ep_rank, self.ep_size = (0, 1) if ep is None else (ep.rank, ep.size)
        self.pack_key = model._pack_key()
        h = C.c_void_p()
        L.check(L.lib().umoe_engine_create(C.byref(c), C.byref(h)), "umoe_engine_create")
        self.h = h
        self.keep: List[torch.Tensor] = []
        self._pack_weights(max_pos)
        if ep is not None and ep.size > 1 and ep_connect:
            ep.connect(self.h)
        self.tokens = None
        self.state = None
        self.io = None
        self.captured = False

    def _k(self, t):
        self.keep.append(t)
        return t

    def _pack_weights(self, max_pos_override=None):
        m, cfg, lib = self.model, self.cfg, L.lib()
        bf = torch.bfloat16
        for p in m.parameters():
            if p.dtype != bf:
                raise L.UmoeError("engine weights must be bfloat16")
        mpk = m.packed()
        self._k(mpk)
        for li, layer in enumerate(m.language_model.layers):
            lp = mpk["layers"][li]
            qkv_w, qkv_b, o_w, pk = lp["qkv_w"], lp["qkv_b"], lp["o_w"], lp["moe"]
            n_real, n_fix = cfg.mlp_dynamic_expert_num, cfg.mlp_fixed_expert_num
            arr = lambda ts: (C.c_void_p * max(1, len(ts)))(*[t.data_ptr() for t in ts])
            # expert parallel: this rank streams its n_real / ep_size local experts (core.py:505) at decode; the prefill runs
            # replicated on the row-major tensors of all experts
            e_loc = n_real // self.ep_size
            # row-major originals (the module's own parameters) for the tiled MFMA kernels of the prefill
            ex = layer.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts
            sh = layer.mlp.fixed_real_moe
            # SHARDED model (from_pretrained(ep_rank, ep_size) / config.ep_size: the modules hold the n_real / ep_size local experts only,
            # core.py:505, deepspeed_ep_param_aggregation.py:16-48): the engine gets exactly those, and no row-major tensors of experts it
            # does not have -- its prefill then runs through the module-level forward (prefill below)
            sharded = self.ep_size > 1 and len(ex) == e_loc
            if not sharded and len(ex) != n_real:
                raise L.UmoeError(f"the model holds {len(ex)} routed experts per layer: expected {n_real} (replicated) or {e_loc} (sharded for ep_size {self.ep_size})")
            self.sharded = sharded
            lo = 0 if sharded else self.ep_rank * e_loc
            eg, ed = arr(pk["exp_gu"][lo:lo + e_loc]), arr(pk["exp_dn"][lo:lo + e_loc])
            sg, sd = arr(pk["sh_gu"]), arr(pk["sh_dn"])
            rm = [p for mods in (ex, sh) for m_ in mods for p in (m_.gate_proj.weight, m_.up_proj.weight, m_.down_proj.weight)]
            if not all(p.is_contiguous() for p in rm + [layer.self_attn.o_proj.weight]):
                raise L.UmoeError("engine weights must be contiguous")
            if sharded:
                reg = reu = red = None
            else:
                reg, reu, red = arr([m_.gate_proj.weight for m_ in ex]), arr([m_.up_proj.weight for m_ in ex]), arr([m_.down_proj.weight for m_ in ex])
            rsg, rsu, rsd = arr([m_.gate_proj.weight for m_ in sh]), arr([m_.up_proj.weight for m_ in sh]), arr([m_.down_proj.weight for m_ in sh])
            w = L.LayerWeights(in_norm=layer.input_layernorm.weight.data_ptr(), qkv_w=qkv_w.data_ptr(), qkv_b=qkv_b.data_ptr(),
                               o_w=o_w.data_ptr(), post_norm=layer.post_attention_layernorm.weight.data_ptr(),
                               gate_w=layer.mlp.gate.weight.data_ptr(), exp_gu=eg, exp_dn=ed, sh_gu=sg, sh_dn=sd,
                               rm_qkv=lp["qkv_rm"].data_ptr(), rm_o=layer.self_attn.o_proj.weight.data_ptr(),
                               rm_exp_gate=reg, rm_exp_up=reu, rm_exp_down=red, rm_sh_gate=rsg, rm_sh_up=rsu, rm_sh_down=rsd)
            L.check(lib.umoe_engine_set_layer(self.h, li, C.byref(w)), "umoe_engine_set_layer")
        emb, head = mpk["emb"], mpk["head"]
        # (slack: the temporal stream of a video advances by seconds-per-grid * tokens_per_second per frame pair, model.py:597-603)
        max_pos = self.Lmax + 4104 if max_pos_override is None else int(max_pos_override)
        self.max_pos = max_pos
        cos, sin = ops.rope_tables(max_pos, cfg.head_dim, cfg.rope_theta, self.dev)
        self._k(cos), self._k(sin)
        delay = (C.c_int32 * cfg.codec_channels)(*cfg.codec_delay_pattern)
        L.check(lib.umoe_engine_set_globals(self.h, m.language_model.norm.weight.data_ptr(), emb.data_ptr(), head.data_ptr(),
                                            cos.data_ptr(), sin.data_ptr(), max_pos, delay), "umoe_engine_set_globals")
        torch.cuda.synchronize()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def prefill(self, x: torch.Tensor, attention_mask: torch.Tensor, position_ids: Optional[torch.Tensor] = None,
                rope_deltas: Optional[torch.Tensor] = None, external: bool = False):
        """position_ids [3, rows, T] / rope_deltas [rows, 1] (get_rope_index): multimodal prompts; a generated token then sits at
        position T + steps + delta on all three streams (the reference's cache_position + rope_deltas, model.py:779-790)."""
        rows, T = attention_mask.shape
        assert rows == self.rows and x.shape == (rows * T, self.cfg.hidden_size) and x.dtype == torch.bfloat16
        valid = attention_mask.to(torch.uint8).cpu().contiguous()
        self.T_prompt = T
        if getattr(self, "sharded", False) or external:
            # (external=True on an engine that COULD prefill itself: the same module-level forward, e.g. as the reference of a sharded run)
            return self._prefill_sharded(x, attention_mask, valid, position_ids, rope_deltas)
        if position_ids is None:
            L.check(L.lib().umoe_engine_prefill(self.h, x.data_ptr(), valid.data_ptr(), T, self._stream()), "umoe_engine_prefill")
        else:
            assert tuple(position_ids.shape) == (3, rows, T) and rope_deltas is not None
            pos = position_ids.to(torch.int32).cpu().contiguous()
            nxt = (rope_deltas.reshape(rows).to(torch.int64).cpu() + T).to(torch.int32).contiguous()
            need = int(max(int(pos.max()), int(nxt.max()) + self.Lmax - T)) + 2
            if need > self.max_pos:
                raise L.UmoeError(f"rope table of the engine covers {self.max_pos} positions, the prompt needs {need}")
            L.check(L.lib().umoe_engine_prefill_pos(self.h, x.data_ptr(), valid.data_ptr(), T, pos.data_ptr(), nxt.data_ptr(), self._stream()),
                    "umoe_engine_prefill_pos")
        self.captured = False

    def _prefill_sharded(self, x, attention_mask, valid, position_ids, rope_deltas):
        """Prefill of a rank that holds ONLY its local experts: the module-level forward (text_forward: the same tiled kernels the engine's
        own prefill runs; its DCMoE blocks exchange the routed rows with the other ranks over the link's process group, core.py:455-488)
        hands every layer's roped K / V to the engine's cache, then the engine takes the decode state (umoe_engine_prefill_external).
        All ranks must call it together (every layer is two collectives)."""
        rows, T = attention_mask.shape
        cfg = self.cfg
        KVH, hd, Lmax = cfg.num_key_value_heads, cfg.head_dim, self.Lmax
        per_layer = rows * KVH * Lmax * hd * 2
        pad_k = torch.zeros((rows, KVH, Lmax, hd), dtype=torch.bfloat16, device=self.dev)
        pad_v = torch.zeros_like(pad_k)

        def sink(li, kc, vc):
            pad_k[:, :, :T] = kc
            pad_v[:, :, :T] = vc
            self.write_buffer("k_cache", pad_k, li * per_layer)
            self.write_buffer("v_cache", pad_v, li * per_layer)

        with torch.no_grad():
            self.model.text_forward(x.view(rows, T, cfg.hidden_size), attention_mask.to(self.dev), position_ids=position_ids, kv_sink=sink)
        pos_p = nxt_p = None
        if position_ids is not None:
            assert tuple(position_ids.shape) == (3, rows, T) and rope_deltas is not None
            pos = position_ids.to(torch.int32).cpu().contiguous()
            nxt = (rope_deltas.reshape(rows).to(torch.int64).cpu() + T).to(torch.int32).contiguous()
            pos_p, nxt_p = pos.data_ptr(), nxt.data_ptr()
        L.check(L.lib().umoe_engine_prefill_external(self.h, valid.data_ptr(), T, pos_p, nxt_p, self._stream()), "umoe_engine_prefill_external")
        self.captured = False

    def start_decode(self, prefill_tokens: torch.Tensor, prefill_steps: List[int], max_tokens: int, min_tokens,
                     cfg_scale, temperature, top_p, top_k, eos_mul, do_sample, seed=0):
        B, Cc = self.batch, self.cfg.codec_channels
        assert prefill_tokens.shape[0] == B and prefill_tokens.shape[2] == Cc
        if max_tokens + 2 > self.Tmax or self.T_prompt + max_tokens + 1 > self.Lmax:
            raise L.UmoeError("engine buffers too small for max_tokens")
        tok = torch.full((B, self.Tmax, Cc), -1, dtype=torch.int32, device=self.dev)
        tok[:, : prefill_tokens.shape[1]] = prefill_tokens.to(self.dev, torch.int32)
        step0 = min(prefill_steps) - 1
        st = torch.zeros(4 * B + 8, dtype=torch.int32)
        st[B:2 * B] = -1                                     # eos_countdown
        st[2 * B:3 * B] = -1                                 # finished_step
        st[3 * B:4 * B] = torch.tensor(prefill_steps, dtype=torch.int32)
        st[4 * B + 0], st[4 * B + 1], st[4 * B + 4] = step0, max_tokens, step0
        self.tokens, self.state = tok, st.to(self.dev)
        self.prefill_steps, self.max_tokens = list(prefill_steps), max_tokens
        self.io = L.DecodeIO(tokens=tok.data_ptr(), state=self.state.data_ptr(), cfg_scale=cfg_scale, temperature=temperature,
                             top_p=top_p, eos_mul=eos_mul, top_k=-1 if top_k is None else int(top_k), do_sample=int(bool(do_sample)),
                             min_tokens=-1 if min_tokens is None else int(min_tokens), seed=seed)
        self.captured = False
        self.steps_run = 0

    def step(self, use_graph: bool = True):
        lib = L.lib()
        if use_graph:
            if not self.captured:
                # capture on a side stream (legacy-stream capture is not allowed), then replay on the current one
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    L.check(lib.umoe_engine_capture(self.h, C.byref(self.io), C.c_void_p(s.cuda_stream)), "umoe_engine_capture")
                torch.cuda.current_stream().wait_stream(s)
                self.captured = True
            L.check(lib.umoe_engine_replay(self.h, self._stream()), "umoe_engine_replay")
        else:
            L.check(lib.umoe_engine_decode_step(self.h, C.byref(self.io), self._stream()), "umoe_engine_decode_step")
        self.steps_run += 1

    KINDS = ["qkv", "rope", "attn", "oproj", "router", "dispatch", "gateup", "down", "combine", "embed", "head", "sample", "delay"]

    def profile_steps(self, n: int = 4) -> dict:
        """n eager steps with HIP events after every kernel class -> {class: (ms per launch incl. gap, launches/step)}."""
        ms = (C.c_float * 13)()
        cnt = (C.c_int * 13)()
        for _ in range(n):
            L.check(L.lib().umoe_engine_profile_step(self.h, C.byref(self.io), self._stream(), ms, cnt, 13), "umoe_engine_profile_step")
            self.steps_run += 1
        return {k: (ms[i] / max(cnt[i], 1), cnt[i] // n) for i, k in enumerate(self.KINDS)}

    def handoff_error(self) -> int:
        """Sticky error word of the in-launch hand-offs (0 = none; 1 = an expert-parallel receive, 2 = the riders' rows, 3 = an expert's
        SwiGLU rows were not published within the bounded wait -- e.g. a workgroup of a fused launch was not resident; 4 / 5 = the
        operand-order row tiles / a local expert's h tiles of the one-launch expert-parallel MoE half, umoe_moe_ep.hip)."""
        return int(self.copy_buffer("ep_words", torch.int32, (2,))[1].item())

    def all_done(self) -> bool:
        done = bool(int(self.state[4 * self.batch + 2].item()))
        code = self.handoff_error()            # (the host is synchronised here anyway: fail loudly instead of decoding on stale rows)
        if code and not (self.ep is not None and code == 1):     # (an EP receive timeout is the caller's to handle: ep_error())
            raise L.UmoeError(f"decode engine: an in-launch hand-off timed out (code {code}); UMOE_RIDER_PUB=0 selects the launch-per-kernel path")
        if self.ep is not None and self.ep.size > 1 and self.ep.mode != "loopback" and self.ep.group is not False:
            # expert parallel: every rank keeps stepping until ALL are done (a rank that stopped would starve its peers' receives)
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized():
                t = torch.tensor([1 if done else 0], dtype=torch.int32)
                if dist.get_backend(self.ep.group) == "nccl":
                    t = t.to(self.dev)
                dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.ep.group)
                done = bool(int(t.item()))
        return done

    def ep_error(self) -> int:
        """0, or 1 when a receive of the expert-parallel exchange timed out (sticky); synchronises the current stream."""
        code = C.c_int()
        L.check(L.lib().umoe_engine_ep_error(self.h, self._stream(), C.byref(code)), "umoe_engine_ep_error")
        return int(code.value)

    def run(self, use_graph: bool = True, poll_every: int = 16, max_steps: Optional[int] = None):
        """Decode until every sequence finished (reference loop head, model.py:1149-1151).  The device decides;
        the host only polls one flag every `poll_every` steps (steps past the end are no-ops on the token state)."""
        budget = self.max_tokens - int(self.state[4 * self.batch].item()) if max_steps is None else max_steps
        done = 0
        while done < budget:
            n = min(poll_every, budget - done)
            for _ in range(n):
                self.step(use_graph)
            done += n
            if self.all_done():
                break
        return done

    def finish(self):
        """reference model.py:1205-1231: lengths, packing of generated_codes."""
        B, cfg = self.batch, self.cfg
        md = max(cfg.codec_delay_pattern)
        st = self.state.cpu()
        dec_step = int(st[4 * B])
        finished = st[2 * B:3 * B].long().clone()
        final_step = dec_step + 1
        finished[finished == -1] = final_step - md
        lengths = torch.clamp(finished - torch.tensor(self.prefill_steps), min=0)
        max_len = int(lengths.max()) + md
        tokens = self.tokens[:, : max(dec_step + 1, 1)].clone()
        if max_len <= 0:
            return None, None, tokens
        out = torch.full((B, max_len, cfg.codec_channels), cfg.codec_pad_value, dtype=torch.long, device=self.dev)
        for i in range(B):
            n = int(lengths[i]) + md
            if n > 0:
                seg = tokens[i, self.prefill_steps[i]: self.prefill_steps[i] + n]
                out[i, : seg.shape[0]] = seg.long()
        return out, lengths.to(self.dev), tokens

    def copy_buffer(self, name: str, dtype: torch.dtype, shape) -> torch.Tensor:
        """Copies an engine workspace buffer into a fresh device tensor (parity tests)."""
        n = C.c_size_t()
        p = L.lib().umoe_engine_buffer(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        out = torch.empty(shape, dtype=dtype, device=self.dev)
        assert out.numel() * out.element_size() <= n.value, (name, n.value)
        torch.cuda.synchronize()
        hip = C.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(p), C.c_size_t(out.numel() * out.element_size()), 3)
        assert rc == 0
        return out

    def info(self, key: str) -> int:
        """Host-side facts about the C engine (umoe_engine_info): "expert_launch" (0 two launches, 1 box-grid fused, 2 flat), "n_cu"."""
        return int(L.lib().umoe_engine_info(self.h, key.encode()))

    def write_buffer(self, name: str, src: torch.Tensor, offset_bytes: int = 0) -> None:
        """Copies a device tensor INTO an engine workspace buffer (parity tests: the oracle's KV cache for the per-layer probe)."""
        n = C.c_size_t()
        p = L.lib().umoe_engine_buffer(self.h, name.encode(), C.byref(n))
        if not p:
            raise KeyError(name)
        src = src.contiguous()
        nb = src.numel() * src.element_size()
        assert src.device == self.dev and offset_bytes >= 0 and offset_bytes + nb <= n.value, (name, n.value, offset_bytes, nb)
        torch.cuda.synchronize()
        hip = C.CDLL("libamdhip64.so")
        rc = hip.hipMemcpy(C.c_void_p(p + offset_bytes), C.c_void_p(src.data_ptr()), C.c_size_t(nb), 3)
        assert rc == 0

    def set_probe(self, teach_x: Optional[torch.Tensor] = None, dump_x1: bool = False, dump_x: bool = False, dump_logits: bool = False):
        """Per-layer probe of the parity tests (umoe_engine_set_probe; eager steps only).  teach_x [layers, rows, D] bf16 replaces the
        residual stream at the start of every layer; the returned dict holds the device tensors the following eager steps fill:
        "x1" / "x" [layers, rows, D] (after attention + o_proj / after the MoE block), "logits" [layers, rows, E] (router, bf16).
        set_probe() with no argument switches the probe off."""
        cfg, dev = self.cfg, self.dev
        Lyr, D, E = cfg.num_hidden_layers, cfg.hidden_size, cfg.num_experts
        out = {}
        if teach_x is not None:
            assert teach_x.shape == (Lyr, self.rows, D) and teach_x.dtype == torch.bfloat16 and teach_x.device == dev
            out["teach"] = teach_x.contiguous()
        if dump_x1:
            out["x1"] = torch.zeros(Lyr, self.rows, D, dtype=torch.bfloat16, device=dev)
        if dump_x:
            out["x"] = torch.zeros(Lyr, self.rows, D, dtype=torch.bfloat16, device=dev)
        if dump_logits:
            out["logits"] = torch.zeros(Lyr, self.rows, E, dtype=torch.bfloat16, device=dev)
        ptr = lambda k: C.c_void_p(out[k].data_ptr()) if k in out else None
        L.check(L.lib().umoe_engine_set_probe(self.h, ptr("teach"), ptr("x1"), ptr("x"), ptr("logits")), "umoe_engine_set_probe")
        self._probe = out            # (keeps the tensors alive while the engine holds their addresses)
        self.captured = False
        return out

    def close(self):
        if getattr(self, "h", None):
            torch.cuda.synchronize()
            L.lib().umoe_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
