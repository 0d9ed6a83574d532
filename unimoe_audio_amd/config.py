"""Host-side mirror of the reference's model hyper-parameters.

Field names follow the reference configs so a reference user finds the same knobs:
`Qwen2_5_VLMoETextConfig` (reference utils/UniMoE_Audio_model.py:66-120) and
`UniAudioRVQQwen2_5VLMoEConfig` (utils/UniMoE_Audio_model.py:123-163); defaults are the
shipped values of reference utils/config.json.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field
from typing import List, Optional


@dataclass
class UniMoEAudioConfig:
    # transformer (config.json:31-51, 126-139)
    hidden_size: int = 2048
    num_hidden_layers: int = 36
    num_attention_heads: int = 16
    num_key_value_heads: int = 2
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1000000.0
    mrope_section: List[int] = field(default_factory=lambda: [16, 24, 24])
    max_position_embeddings: int = 128000
    vocab_size: int = 151676
    hidden_act: str = "silu"
    initializer_range: float = 0.02
    # DCMoE (config.json:58-77, 118-143)
    mlp_dynamic_expert_num: int = 8
    mlp_dynamic_null_expert_num: int = 1
    mlp_dynamic_top_p: float = 0.7
    mlp_dynamic_top_k: int = 0
    mlp_fixed_expert_num: int = 2
    dynamic_intermediate_size: int = 2752
    shared_intermediate_size: int = 1376
    ignore_differentiable_router: bool = True
    enable_expert_tensor_parallelism: bool = False
    ep_size: int = 1
    fixed_ep_size: int = 1
    router_jitter_noise: float = 0.01
    input_jitter_noise: float = 0.01
    token_drop: bool = False
    drop_policy: str = "probs"
    min_capacity: int = 8
    capacity_factor: float = 6.0
    fp32_gate: bool = True
    avg_hidden_states_last: bool = False
    drop_token_num_print: bool = True
    l_aux_weight: float = 0.025
    min_l_aux_weight: float = 0.001
    l_aux_weight_decay_steps: int = 10000
    # codec (config.json:7-26)
    codec_vocab_size: int = 1027
    codec_delay_pattern: List[int] = field(default_factory=lambda: [0, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
    codec_channels: int = 12
    codec_eos_value: int = 1024
    codec_pad_value: int = 1025
    codec_bos_value: int = 1026
    codec_placeholder_value: Optional[int] = 151665
    # multimodal (config.json:32,147-185): token ids and the vision tower's geometry (None = text / audio only)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_token_id: int = 151652
    vision_end_token_id: int = 151653
    vision_config: Optional[dict] = None

    # ---- derived -------------------------------------------------------------
    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def num_dyn(self) -> int:
        """dynamic router columns incl. null experts (reference core.py:205)"""
        return self.mlp_dynamic_expert_num + self.mlp_dynamic_null_expert_num

    @property
    def num_experts(self) -> int:
        """router columns: dynamic + null + shared (reference core.py:211)"""
        return self.num_dyn + self.mlp_fixed_expert_num

    @property
    def text_config(self):  # the reference nests the text config; expose the same spelling
        return self

    def to_dict(self):
        return asdict(self)

    @classmethod
    def from_json(cls, path: str) -> "UniMoEAudioConfig":
        """Reads a reference-format config.json (top-level codec_* keys + nested text_config)."""
        with open(path) as f:
            raw = json.load(f)
        text = dict(raw.get("text_config", {}))
        merged = {**text, **{k: v for k, v in raw.items() if k.startswith("codec_")}}
        rope = text.get("rope_scaling") or raw.get("rope_scaling") or {}
        if "mrope_section" in rope:
            merged["mrope_section"] = rope["mrope_section"]
        for k in ("image_token_id", "video_token_id", "vision_start_token_id", "vision_end_token_id", "vision_config"):
            if raw.get(k) is not None:
                merged[k] = raw[k]
        names = {f.name for f in cls.__dataclass_fields__.values()}
        kw = {k: v for k, v in merged.items() if k in names}
        if "mlp_dynamic_top_k" in kw:
            kw["mlp_dynamic_top_k"] = int(kw["mlp_dynamic_top_k"])
        return cls(**kw)

    @classmethod
    def tiny(cls, **over) -> "UniMoEAudioConfig":
        """Small shapes for CPU-speed parity tests (same structure, same flags)."""
        kw = dict(
            hidden_size=128,
            num_hidden_layers=2,
            num_attention_heads=4,
            num_key_value_heads=2,
            mrope_section=[4, 6, 6],
            vocab_size=320,
            dynamic_intermediate_size=96,
            shared_intermediate_size=64,
            codec_vocab_size=1027,
            codec_placeholder_value=300,
        )
        kw.update(over)
        return cls(**kw)
