"""CPU restatement of the reference vision tower and of the 3-D mRoPE index (torch-CPU).

TEST INFRASTRUCTURE ONLY: imported by tests/ -- never by the product package.

Follows the reference (read as text):
  Conv3D patch embedding = linear map of the flattened [C, Tp, P, P] patch      utils/UniMoE_Audio_utils.py:585-725
  rot_pos_emb (2-D rotary table indexed by (h, w) of every patch, merge order)  utils/UniMoE_Audio_utils.py:786-813
  get_window_index (merged tokens regrouped window by window)                   utils/UniMoE_Audio_utils.py:815-854
  forward: permute to window order, 32 blocks (window / full attention by cu_seqlens), merger, un-permute   :856-900
  block / attention / MLP / merger arithmetic: third-party transformers==4.53.1 `Qwen2_5_VLVisionBlock`, `Qwen2_5_VLVisionAttention`
      (qkv Linear with bias, rotary applied in fp32 and cast back, NON-causal attention inside each cu_seqlens segment, proj),
      `Qwen2_5_VLMLP` (down(silu(gate x) * up x), biases), `Qwen2_5_VLPatchMerger` (RMSNorm, 4 tokens concatenated, Linear-GELU-Linear)
  get_rope_index                                                                utils/UniMoE_Audio_model.py:513-652
Parity pin: tests/golden/vision_tower.npz, rope_index.npz (the reference's own classes, oracle/gen_golden.py::gen_vision).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


def rmsnorm(x, w, eps=1e-6):
    xf = x.float()
    return w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(x.dtype)


def rot_pos_emb(grid_thw: torch.Tensor, head_dim: int, merge: int, theta: float = 10000.0) -> torch.Tensor:
    """[n_patches, head_dim / 2] fp32: (h, w) rotary angles of every patch, patches in merge-block order (utils.py:786-813)"""
    pos = []
    for t, h, w in grid_thw.tolist():
        hp = torch.arange(h).unsqueeze(1).expand(-1, w).reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        wp = torch.arange(w).unsqueeze(0).expand(h, -1).reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).flatten()
        pos.append(torch.stack([hp, wp], dim=-1).repeat(t, 1))
    pos = torch.cat(pos, 0)
    dim = head_dim // 2
    inv = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
    table = torch.outer(torch.arange(int(grid_thw[:, 1:].max()), dtype=torch.float), inv)
    return table[pos].flatten(1)


def window_index(grid_thw: torch.Tensor, window_size: int, merge: int, patch: int) -> Tuple[torch.Tensor, List[int]]:
    """utils.py:815-854"""
    idx_all, cu = [], [0]
    base = 0
    ws = window_size // merge // patch
    for t, h, w in grid_thw.tolist():
        gh, gw = h // merge, w // merge
        index = torch.arange(t * gh * gw).reshape(t, gh, gw)
        ph, pw = ws - gh % ws, ws - gw % ws
        nh, nw = (gh + ph) // ws, (gw + pw) // ws
        padded = F.pad(index, (0, pw, 0, ph), "constant", -100).reshape(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
        lens = (padded != -100).sum([2, 3]).reshape(-1)
        flat = padded.reshape(-1)
        idx_all.append(flat[flat != -100] + base)
        cu.extend((lens.cumsum(0) * merge * merge + cu[-1]).tolist())
        base += t * gh * gw
    return torch.cat(idx_all, 0), cu


def vision_forward(cfg: dict, w: Dict[str, torch.Tensor], x: torch.Tensor, grid_thw: torch.Tensor) -> torch.Tensor:
    """x [n_patches, C*Tp*P*P] -> [n_patches / merge^2, out_hidden]"""
    H, hidden, merge = cfg["num_heads"], cfg["hidden_size"], cfg["spatial_merge_size"]
    hd, unit = hidden // H, merge * merge
    h = F.linear(x, w["patch_embed.proj.weight"].reshape(hidden, -1))                       # Conv3D == linear on the flat patch
    rot = rot_pos_emb(grid_thw, hd, merge)
    widx, cu_win = window_index(grid_thw, cfg["window_size"], merge, cfg["patch_size"])
    cu_win = torch.unique_consecutive(torch.tensor(cu_win, dtype=torch.int32)).tolist()
    S = h.shape[0]
    h = h.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
    rot = rot.reshape(S // unit, unit, -1)[widx].reshape(S, -1)
    emb = torch.cat((rot, rot), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    cu_full = F.pad(torch.repeat_interleave(grid_thw[:, 1] * grid_thw[:, 2], grid_thw[:, 0]).cumsum(0), (1, 0), value=0).tolist()

    def rot_half(t):
        return torch.cat((-t[..., t.shape[-1] // 2:], t[..., : t.shape[-1] // 2]), dim=-1)

    for l in range(cfg["depth"]):
        p = f"blocks.{l}."
        cu = cu_full if l in cfg["fullatt_block_indexes"] else cu_win
        y = rmsnorm(h, w[p + "norm1.weight"])
        q, k, v = F.linear(y, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"]).reshape(S, 3, H, hd).permute(1, 0, 2, 3).unbind(0)
        qf, kf = q.float(), k.float()
        c, s_ = cos.unsqueeze(-2).float(), sin.unsqueeze(-2).float()
        q = (qf * c + rot_half(qf) * s_).to(y.dtype)
        k = (kf * c + rot_half(kf) * s_).to(y.dtype)
        outs = []
        for a, b in zip(cu[:-1], cu[1:]):
            qs, ks, vs = (t[a:b].transpose(0, 1).unsqueeze(0) for t in (q, k, v))
            outs.append(F.scaled_dot_product_attention(qs, ks, vs, is_causal=False)[0].transpose(0, 1))
        ao = torch.cat(outs, 0).reshape(S, -1)
        h = h + F.linear(ao, w[p + "attn.proj.weight"], w[p + "attn.proj.bias"])
        y = rmsnorm(h, w[p + "norm2.weight"])
        g = F.linear(y, w[p + "mlp.gate_proj.weight"], w[p + "mlp.gate_proj.bias"])
        u = F.linear(y, w[p + "mlp.up_proj.weight"], w[p + "mlp.up_proj.bias"])
        h = h + F.linear(F.silu(g) * u, w[p + "mlp.down_proj.weight"], w[p + "mlp.down_proj.bias"])
    z = rmsnorm(h, w["merger.ln_q.weight"]).view(-1, hidden * unit)
    z = F.linear(F.gelu(F.linear(z, w["merger.mlp.0.weight"], w["merger.mlp.0.bias"])), w["merger.mlp.2.weight"], w["merger.mlp.2.bias"])
    return z[torch.argsort(widx)]


def rope_index(input_ids, image_grid_thw, video_grid_thw, second_per_grid_ts, attention_mask, *, merge: int, tokens_per_second: float,
               image_token_id: int, video_token_id: int, vision_start_token_id: int):
    """model.py:513-652 -> (position_ids [3, B, T], deltas [B, 1])"""
    if input_ids is None or (image_grid_thw is None and video_grid_thw is None):
        if attention_mask is not None:
            pos = attention_mask.long().cumsum(-1) - 1
            pos = pos.masked_fill(attention_mask == 0, 1).unsqueeze(0).expand(3, -1, -1)
            mx = pos.max(0)[0].max(-1, keepdim=True)[0]
            return pos, mx + 1 - attention_mask.shape[-1]
        B, T = input_ids.shape
        return torch.arange(T).view(1, 1, -1).expand(3, B, -1), torch.zeros((B, 1), dtype=input_ids.dtype)
    if attention_mask is None:
        attention_mask = torch.ones_like(input_ids)
    pos_all = torch.ones(3, *input_ids.shape, dtype=input_ids.dtype)
    deltas = []
    ii = vi = 0
    for r in range(input_ids.shape[0]):
        toks = input_ids[r][attention_mask[r] == 1].tolist()
        starts = [i for i, t in enumerate(toks) if t == vision_start_token_id]
        n_img = sum(1 for i in starts if toks[i + 1] == image_token_id)
        n_vid = sum(1 for i in starts if toks[i + 1] == video_token_id)
        pieces, st = [], 0
        for _ in range(n_img + n_vid):
            e_img = toks.index(image_token_id, st) if (image_token_id in toks and n_img > 0) else len(toks) + 1
            e_vid = toks.index(video_token_id, st) if (video_token_id in toks and n_vid > 0) else len(toks) + 1
            if e_img < e_vid:
                t, h, w = image_grid_thw[ii].tolist()
                sec, ii, n_img, ed = 0, ii + 1, n_img - 1, e_img
            else:
                t, h, w = video_grid_thw[vi].tolist()
                sec = second_per_grid_ts[vi] if second_per_grid_ts is not None else 1.0
                vi, n_vid, ed = vi + 1, n_vid - 1, e_vid
            gh, gw = h // merge, w // merge
            text_len = ed - st
            base = int(pieces[-1].max()) + 1 if pieces else 0
            pieces.append(torch.arange(text_len).view(1, -1).expand(3, -1) + base)
            # `second_per_grid_t` is cast to the LONG dtype of the frame index before the product (model.py:597-601): 2.5 s -> 2, 0.5 s -> 0
            tt = (torch.arange(t).view(-1, 1).expand(-1, gh * gw) * torch.as_tensor(sec, dtype=torch.long) * tokens_per_second).long().flatten()
            hh = torch.arange(gh).view(1, -1, 1).expand(t, -1, gw).flatten()
            ww = torch.arange(gw).view(1, 1, -1).expand(t, gh, -1).flatten()
            pieces.append(torch.stack([tt, hh, ww]) + text_len + base)
            st = ed + t * gh * gw
        if st < len(toks):
            base = int(pieces[-1].max()) + 1 if pieces else 0
            pieces.append(torch.arange(len(toks) - st).view(1, -1).expand(3, -1) + base)
        llm = torch.cat(pieces, dim=1).reshape(3, -1)
        pos_all[:, r, attention_mask[r] == 1] = llm.to(pos_all.dtype)
        deltas.append(int(llm.max()) + 1 - input_ids.shape[1])
    return pos_all, torch.tensor(deltas).unsqueeze(1)
