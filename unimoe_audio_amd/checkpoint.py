"""Checkpoint I/O for the accelerated path: Hugging Face safetensors shards in, bf16 device parameters out.

What the reference does:
  * `from_pretrained(local_dir, torch_dtype=bf16, ...)` over sharded `*.safetensors` + `model.safetensors.index.json`
    (README.md:64-87, utils/UniMoE_Audio_mod.py:79-92);
  * key conversion `_checkpoint_conversion_mapping` (utils/UniMoE_Audio_model.py:464-467): `^visual` stays, every other `model.*`
    key that is not already `model.language_model.*` / `model.visual.*` moves under `language_model.`;
  * expert parallelism: rank r of `ep_size` owns the routed experts [r * E / ep_size, (r + 1) * E / ep_size) under LOCAL indices
    (utils/UniMoE_Audio_core.py:505), and UniMoEV2-Preview/inference/deepspeed_ep_param_aggregation.py:16-48 re-groups expert
    tensors between EP degrees: global expert e -> rank e // (E / ep_size), local id e % (E / ep_size), everything else replicated,
    one file per rank named `model-expert_{i}-of-total_{T}.safetensors`.

Here: shards are streamed tensor by tensor (safetensors is memory mapped: no full host copy of the 14 GB model), cast to the
parameter's dtype and copied straight into the parameter that already lives on its device.  Nothing in this file computes.
"""
from __future__ import annotations

import glob
import json
import os
import re
from typing import Dict, Iterable, Iterator, List, Optional, Tuple

import torch

INDEX_NAME = "model.safetensors.index.json"
_EXPERT_RE = re.compile(r"^(.*\.mlp\.dynamic_real_moe\.deepspeed_moe\.experts\.deepspeed_experts\.)(\d+)(\..*)$")
_EP_FILE_RE = re.compile(r"^model-expert_(\d+)-of-total_(\d+)\.safetensors$")
# tensors of the checkpoint that the accelerated path does not own (vision tower: SURVEY 8f-1; lm_head: unused by the codec loss)
COLD_PREFIXES = ("visual.", "lm_head.")


def convert_key(key: str) -> str:
    """The reference's `_checkpoint_conversion_mapping` (model.py:464-467)."""
    if key.startswith("visual"):
        return key
    if key.startswith("model.language_model.") or key.startswith("model.visual."):
        return key[len("model."):]
    if key.startswith("model"):
        return "language_model" + key[len("model"):]
    return key


def ep_owner(expert: int, n_experts: int, ep_size: int) -> Tuple[int, int]:
    """global routed expert -> (rank, local index); deepspeed_ep_param_aggregation.py:18-22,38."""
    if ep_size < 1 or n_experts % ep_size:
        raise ValueError(f"ep_size {ep_size} must divide the {n_experts} routed experts")
    per = n_experts // ep_size
    return expert // per, expert % per


def ep_local_key(key: str, n_experts: int, ep_rank: int, ep_size: int) -> Optional[str]:
    """Key of a FULL checkpoint as rank `ep_rank` stores it: routed experts of other ranks -> None, own experts renumbered."""
    m = _EXPERT_RE.match(key)
    if not m or ep_size == 1:
        return key
    rank, local = ep_owner(int(m.group(2)), n_experts, ep_size)
    return f"{m.group(1)}{local}{m.group(3)}" if rank == ep_rank else None


def _shards(path: str) -> List[Tuple[str, Optional[List[str]]]]:
    """[(shard file, keys to read or None = all)] in a deterministic order: the HF index when present, else every shard."""
    idx = os.path.join(path, INDEX_NAME)
    if os.path.exists(idx):
        with open(idx) as f:
            weight_map: Dict[str, str] = json.load(f)["weight_map"]
        by_file: Dict[str, List[str]] = {}
        for k, fn in weight_map.items():
            by_file.setdefault(fn, []).append(k)
        missing = [fn for fn in by_file if not os.path.exists(os.path.join(path, fn))]
        if missing:
            raise FileNotFoundError(f"{INDEX_NAME} names shards that are not in {path!r}: {sorted(missing)[:3]}")
        return [(os.path.join(path, fn), sorted(ks)) for fn, ks in sorted(by_file.items())]
    files = sorted(f for f in glob.glob(os.path.join(path, "*.safetensors")) if not _EP_FILE_RE.match(os.path.basename(f)))
    return [(f, None) for f in files]


def iter_tensors(path: str, ep_rank: int = 0, ep_size: int = 1) -> Iterator[Tuple[str, "torch.Tensor"]]:
    """(checkpoint key, host tensor) of every tensor rank `ep_rank` needs; per-rank EP files win when their degree matches."""
    from safetensors import safe_open
    ep_files = {int(m.group(1)): f for f in os.listdir(path) if (m := _EP_FILE_RE.match(f)) and int(m.group(2)) == ep_size} \
        if os.path.isdir(path) else {}
    if ep_size > 1 and len(ep_files) == ep_size:
        with safe_open(os.path.join(path, ep_files[ep_rank]), framework="pt") as f:      # already local expert ids
            for k in sorted(f.keys()):
                yield "local:" + k, f.get_tensor(k)
        return
    shards = _shards(path)
    if not shards:
        raise FileNotFoundError(f"no *.safetensors under {path!r} (the reference downloads them from the HF hub)")
    for fn, keys in shards:
        with safe_open(fn, framework="pt") as f:
            for k in (keys if keys is not None else sorted(f.keys())):
                yield k, f.get_tensor(k)


@torch.no_grad()
def load_checkpoint(model: torch.nn.Module, path: str, ep_rank: int = 0, ep_size: int = 1, strict_hot: bool = True) -> Tuple[List[str], List[str]]:
    """Stream a reference checkpoint directory into `model` (parameters keep their device and dtype).
    Returns (missing, unexpected) like `load_state_dict(strict=False)`; with `strict_hot` a missing hot-path tensor is a KeyError
    and a shape mismatch a ValueError.  Cold tensors (vision tower, lm_head) may be absent on either side."""
    target = dict(model.state_dict(keep_vars=True))
    n_experts = int(getattr(model.config, "mlp_dynamic_expert_num", 0))
    seen, unexpected = set(), []
    for raw, t in iter_tensors(path, ep_rank, ep_size):
        if raw.startswith("local:"):
            key = convert_key(raw[len("local:"):])
        else:
            key = ep_local_key(convert_key(raw), n_experts, ep_rank, ep_size)
            if key is None:
                continue                                   # another rank's routed expert
        dst = target.get(key)
        if dst is None:
            unexpected.append(key)
            continue
        if tuple(dst.shape) != tuple(t.shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(t.shape)} != parameter shape {tuple(dst.shape)}")
        dst.copy_(t.to(dst.dtype), non_blocking=False)     # (not `.data.copy_`: the in-place op must bump `_version`)
        seen.add(key)
    invalidate_packed(model)
    missing = [k for k in target if k not in seen]
    hot = [k for k in missing if not k.startswith(COLD_PREFIXES)]
    if strict_hot and hot:
        raise KeyError(f"checkpoint {path!r} lacks {len(hot)} hot-path tensors, e.g. {hot[:4]}")
    return missing, [k for k in unexpected if not k.startswith(COLD_PREFIXES)]


def init_moe_from_dense(dense_mlp: Dict[str, "torch.Tensor"], target_shapes: Dict[str, tuple], *, moe_copy: str = "all",
                        n_dynamic_experts: int = 8, ep_rank: int = 0, ep_size: int = 1, initializer_range: float = 0.02,
                        generator: Optional["torch.Generator"] = None) -> Dict[str, "torch.Tensor"]:
    """Dense -> MoE FFN initialisation of the reference's training entry (UniMoEV2-Preview/training/train_unimoev2_qwen2vl.py:
    155-240): every expert tensor `...layers.L.mlp.(fixed_real_moe|dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts).E.<rest>`
    is cut out of the dense `...layers.L.mlp.<rest>` -- rows [offset, offset + I) of gate / up (output-feature slices), columns of
    down -- with ONE running offset per source tensor that advances over the dynamic experts in name order and wraps (asserting it
    lands exactly on the end), while shared experts always take the slice at offset 0.  `moe_copy`: "all" = every expert is cut /
    copied; "single" = expert 0 only, the others N(0, initializer_range^2) weights / zero biases; "none" = nothing.  Expert-parallel
    quirk kept: the ROW path starts rank r at (r * experts_per_rank * I) % rows, the COLUMN path always at 0 (:186-189 vs :209-213).
    dense_mlp: {"<prefix>.layers.L.mlp.gate_proj.weight": tensor, ...}; target_shapes: name -> shape of every expert tensor of the
    MoE model.  Returns name -> tensor for the expert tensors it initialises."""
    import re
    out: Dict[str, torch.Tensor] = {}
    if moe_copy == "none":
        return out
    pat = re.compile(r"(.*layers\.(\d+)\.mlp)\.(fixed_real_moe|dynamic_real_moe\.deepspeed_moe\.experts\.deepspeed_experts)\.(\d+)(\..*)")
    offset: Dict[str, int] = {}
    per_rank = n_dynamic_experts // ep_size
    for name, shape in target_shapes.items():
        m = pat.match(name)
        if not m:
            continue
        src_name = m.group(1) + m.group(5)
        kind, expert = m.group(3), int(m.group(4))
        src = dense_mlp[src_name]
        shared = kind == "fixed_real_moe"
        if moe_copy == "all" or expert == 0:
            if shape[0] != src.shape[0]:
                if not shared and src_name not in offset:
                    offset[src_name] = (ep_rank * per_rank * shape[0]) % src.shape[0]
                off = 0 if shared else offset[src_name]
                out[name] = src[off: off + shape[0]].clone()
                if not shared:
                    offset[src_name] += shape[0]
                    if offset[src_name] >= src.shape[0]:
                        assert offset[src_name] == src.shape[0], (src.shape[0], shape[0], offset[src_name])
                        offset[src_name] %= src.shape[0]
            elif len(shape) > 1 and shape[1] != src.shape[1]:
                if not shared and src_name not in offset:
                    offset[src_name] = 0
                off = 0 if shared else offset[src_name]
                out[name] = src[:, off: off + shape[1]].clone()
                if not shared:
                    offset[src_name] += shape[1]
                    if offset[src_name] >= src.shape[1]:
                        assert offset[src_name] == src.shape[1], (src.shape[1], shape[1], offset[src_name])
                        offset[src_name] %= src.shape[1]
            else:
                out[name] = src.clone()
        elif moe_copy == "single":
            if name.endswith("weight"):
                out[name] = torch.empty(tuple(shape), dtype=src.dtype).normal_(mean=0.0, std=initializer_range, generator=generator)
            else:
                out[name] = torch.zeros(tuple(shape), dtype=src.dtype)
    return out


def invalidate_packed(model: torch.nn.Module) -> None:
    """Drop every MFMA-packed weight copy derived from the parameters (model.packed(), each DCMoE block's prepare()) and the
    decode engine built on them: the next forward / generate re-packs from the live tensors.  Called after every load; call it
    yourself after writing parameters through `.data` (which does not bump `_version`, the caches' key)."""
    if hasattr(model, "_pk"):
        model._pk, model._pk_key = None, None
    for m in model.modules():
        if hasattr(m, "_packed"):
            m._packed, m._packed_key = None, None
    eng = getattr(model, "_engine", None)
    if eng is not None:
        eng.close()
        model._engine = None


def save_checkpoint(state: Dict[str, "torch.Tensor"], path: str, max_shard_bytes: int = 4 << 30, reference_keys: bool = True) -> List[str]:
    """Write HF-style shards `model-0000i-of-0000n.safetensors` + index.  `reference_keys`: store `language_model.*` under the
    reference's on-disk spelling `model.*` (what `convert_key` undoes).  Returns the shard file names."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    items = []
    for k, v in state.items():
        kk = "model" + k[len("language_model"):] if reference_keys and k.startswith("language_model.") else k
        items.append((kk, v.detach().to("cpu").contiguous()))
    groups: List[List[Tuple[str, torch.Tensor]]] = [[]]
    size = 0
    for k, v in items:
        nb = v.numel() * v.element_size()
        if groups[-1] and size + nb > max_shard_bytes:
            groups.append([])
            size = 0
        groups[-1].append((k, v))
        size += nb
    names, weight_map, total = [], {}, 0
    for i, g in enumerate(groups):
        fn = f"model-{i + 1:05d}-of-{len(groups):05d}.safetensors"
        save_file(dict(g), os.path.join(path, fn), metadata={"format": "pt"})
        names.append(fn)
        for k, v in g:
            weight_map[k] = fn
            total += v.numel() * v.element_size()
    with open(os.path.join(path, INDEX_NAME), "w") as f:
        json.dump({"metadata": {"total_size": total}, "weight_map": weight_map}, f, indent=1, sort_keys=True)
    return names


def reshard_experts(path_in: str, path_out: str, n_experts: int, target_ep_size: int) -> List[str]:
    """Full checkpoint -> one file per expert-parallel rank (`model-expert_{i}-of-total_{T}.safetensors`: routed experts of the
    rank under local ids, everything else replicated), the layout deepspeed_ep_param_aggregation.py:42-44 writes."""
    from safetensors.torch import save_file
    os.makedirs(path_out, exist_ok=True)
    outs = [dict() for _ in range(target_ep_size)]
    for raw, t in iter_tensors(path_in):
        m = _EXPERT_RE.match(raw)
        if m:
            rank, local = ep_owner(int(m.group(2)), n_experts, target_ep_size)
            outs[rank][f"{m.group(1)}{local}{m.group(3)}"] = t
        else:
            for o in outs:
                o[raw] = t
    names = []
    for i, o in enumerate(outs):
        fn = f"model-expert_{i}-of-total_{target_ep_size}.safetensors"
        save_file(o, os.path.join(path_out, fn), metadata={"format": "pt"})
        names.append(fn)
    return names


def from_pretrained(cls, local_dir: str, torch_dtype=torch.bfloat16, attn_implementation: Optional[str] = None, device=None,
                    ep_rank: int = 0, ep_size: int = 1, config=None):
    """`Model.from_pretrained(local_dir, torch_dtype=bf16, attn_implementation="sdpa"|"eager")` of the reference
    (utils/UniMoE_Audio_mod.py:79-92) for LOCAL directories (the reference never passes anything else: UniMoE_Audio.py:24,60-65).
    attn_implementation is accepted for signature compatibility: attention always runs on the HIP kernels."""
    from .config import UniMoEAudioConfig
    if attn_implementation not in (None, "sdpa", "eager", "flash_attention_2"):
        raise ValueError(f"unknown attn_implementation {attn_implementation!r}")
    if torch_dtype not in (torch.bfloat16, None):
        raise ValueError("the accelerated path computes in bf16 only (the reference loads bf16 too: mod.py:84)")
    if config is None:
        cj = os.path.join(local_dir, "config.json")
        if not os.path.exists(cj):
            raise FileNotFoundError(f"{local_dir!r} must contain the reference config.json")
        config = UniMoEAudioConfig.from_json(cj)
    if ep_size != 1:
        config.ep_size = ep_size
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        if device is not None:
            with torch.device(device):
                model = cls(config)
        else:
            model = cls(config)
    finally:
        torch.set_default_dtype(old)
    load_checkpoint(model, local_dir, ep_rank=ep_rank, ep_size=ep_size)
    return model.eval()
