"""Vision tower (Qwen2.5-VL ViT) and the multimodal prompt glue on the HIP path (BASELINE configs[4]).

Mirrors the reference's `Qwen2_5_VisionTransformerPretrainedModel` (utils/UniMoE_Audio_utils.py:756-900: patch embedding :703-725,
rot_pos_emb :786-813, get_window_index :815-854, forward :856-900) -- same constructor config, same parameter names
(`patch_embed.proj.weight`, `blocks.N.{norm1,norm2,attn.qkv,attn.proj,mlp.gate_proj,mlp.up_proj,mlp.down_proj}`,
`merger.{ln_q,mlp.0,mlp.2}`), so reference checkpoints load by name -- and `get_rope_index` / the video-embedding scatter of the model
(utils/UniMoE_Audio_model.py:513-652,708-751).  Projections run on umoe_tiled_gemm (row-major weights, bias / residual epilogues),
the rest on the kernels of csrc/umoe_vision.hip; integer tables (window order, rotary positions, 3-D mRoPE index) are host-side
index arithmetic, re-derived in vectorised form.  No CPU path: CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# ----------------------------------------------------------------------------------------------- integer tables (host)
def rot_pos_ids(grid_thw: torch.Tensor, merge: int) -> torch.Tensor:
    """[n_patches, 2] (h, w) of every patch in the order the processor emits them: merge x merge blocks, row-major inside a block
    (reference rot_pos_emb, utils.py:786-808)."""
    out = []
    for t, h, w in grid_thw.tolist():
        bh, bw = torch.meshgrid(torch.arange(h // merge), torch.arange(w // merge), indexing="ij")
        ih, iw = torch.meshgrid(torch.arange(merge), torch.arange(merge), indexing="ij")
        hp = (bh[:, :, None, None] * merge + ih[None, None]).reshape(-1)
        wp = (bw[:, :, None, None] * merge + iw[None, None]).reshape(-1)
        out.append(torch.stack([hp, wp], -1).repeat(t, 1))
    return torch.cat(out, 0)


def window_index(grid_thw: torch.Tensor, window_size: int, merge: int, patch: int) -> Tuple[torch.Tensor, List[int]]:
    """Order of the merged tokens window by window, and the cumulative PATCH counts of the windows (reference get_window_index,
    utils.py:815-854).  A window = `ws x ws` merged tokens of one temporal slice, ws = window_size / merge / patch; the grid is
    padded up to the NEXT multiple of ws (one whole extra window row / column when it already divides: the reference's
    `ws - g % ws`), empty windows contribute zero-length segments."""
    ws = window_size // merge // patch
    order, cu, base = [], [0], 0
    for t, h, w in grid_thw.tolist():
        gh, gw = h // merge, w // merge
        nh, nw = (gh + ws - gh % ws) // ws, (gw + ws - gw % ws) // ws
        ids = torch.arange(t * gh * gw).reshape(t, gh, gw)
        for ti in range(t):
            for a in range(nh):
                for b in range(nw):
                    blk = ids[ti, a * ws:(a + 1) * ws, b * ws:(b + 1) * ws].reshape(-1)
                    order.append(blk + base)
                    cu.append(cu[-1] + blk.numel() * merge * merge)
        base += t * gh * gw
    return torch.cat(order, 0), cu


def rope_index(input_ids: torch.Tensor, image_grid_thw, video_grid_thw, second_per_grid_ts, attention_mask, *, spatial_merge_size: int,
               tokens_per_second: float, image_token_id: int, video_token_id: int, vision_start_token_id: int):
    """3-D mRoPE positions of a multimodal prompt (reference get_rope_index, model.py:513-652): text tokens advance all three streams
    together; the tokens of an image / video take (t, h, w) grid coordinates offset by the running position, t scaled by
    second_per_grid_t * tokens_per_second for videos; after a vision span the text resumes at max + 1.  Returns
    (position_ids [3, B, T] with 1 on padded slots, rope_deltas [B, 1] = max position + 1 - T)."""
    if input_ids is None or (image_grid_thw is None and video_grid_thw is None):
        am = attention_mask if attention_mask is not None else torch.ones_like(input_ids)
        pos = (am.long().cumsum(-1) - 1).masked_fill(am == 0, 1)
        if attention_mask is None:
            pos = torch.arange(input_ids.shape[1]).expand(input_ids.shape[0], -1)
            return pos[None].expand(3, -1, -1), torch.zeros((input_ids.shape[0], 1), dtype=input_ids.dtype)
        return pos[None].expand(3, -1, -1), pos.max(-1, keepdim=True)[0] + 1 - am.shape[-1]
    ids_c, am_c = input_ids.cpu(), (attention_mask if attention_mask is not None else torch.ones_like(input_ids)).cpu()
    B, T = ids_c.shape
    pos = torch.ones((3, B, T), dtype=ids_c.dtype)
    deltas = []
    img_i = vid_i = 0
    for r in range(B):
        valid = am_c[r] == 1
        toks = ids_c[r][valid]
        n = toks.numel()
        is_vis = (toks == image_token_id) | (toks == video_token_id)
        # a vision span starts at a vision token whose predecessor is not the same kind of token
        prev = torch.cat([torch.tensor([-1]), toks[:-1]])
        starts = torch.nonzero(is_vis & (prev != toks)).flatten().tolist()
        row = torch.empty((3, n), dtype=ids_c.dtype)
        cur, nxt = 0, 0                                  # next token index to place, next free position value
        for st in starts:
            if st < cur:
                continue                                 # inside a span already placed
            span = st - cur
            row[:, cur:st] = torch.arange(span)[None] + nxt
            nxt += span
            if int(toks[st]) == image_token_id:
                t, h, w = (int(v) for v in image_grid_thw[img_i])
                sec, img_i = 0.0, img_i + 1
            else:
                t, h, w = (int(v) for v in video_grid_thw[vid_i])
                sec = float(second_per_grid_ts[vid_i]) if second_per_grid_ts is not None else 1.0
                vid_i += 1
            gh, gw = h // spatial_merge_size, w // spatial_merge_size
            m = t * gh * gw
            # the reference casts the seconds to the LONG dtype of its index tensor before multiplying (model.py:597-601): 0.5 s -> 0
            tt = (torch.arange(t) * int(sec) * tokens_per_second).long()
            row[0, st:st + m] = tt.repeat_interleave(gh * gw) + nxt
            row[1, st:st + m] = torch.arange(gh).repeat_interleave(gw).repeat(t) + nxt
            row[2, st:st + m] = torch.arange(gw).repeat(t * gh) + nxt
            nxt = int(row[:, st:st + m].max()) + 1
            cur = st + m
        if cur < n:
            row[:, cur:] = torch.arange(n - cur)[None] + nxt
        pos[:, r, valid] = row
        deltas.append(int(row.max()) + 1 - T)
    return pos.to(input_ids.device), torch.tensor(deltas, dtype=ids_c.dtype, device=input_ids.device).unsqueeze(1)


# ----------------------------------------------------------------------------------------------- parameter containers
class _RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.variance_epsilon = eps


class _PatchEmbed(nn.Module):
    """reference Qwen2_5_VisionPatchEmbed / Conv3D (utils.py:585-725): weight [embed, in_ch * temporal_patch, patch, patch], no bias"""

    def __init__(self, patch, tpatch, in_ch, dim):
        super().__init__()
        self.proj = nn.Module()
        self.proj.weight = nn.Parameter(torch.empty(dim, in_ch * tpatch, patch, patch))
        self.proj.register_parameter("bias", None)


class _VisionAttn(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=True)
        self.proj = nn.Linear(dim, dim, bias=True)


class _VisionMLP(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.gate_proj = nn.Linear(dim, inter, bias=True)
        self.up_proj = nn.Linear(dim, inter, bias=True)
        self.down_proj = nn.Linear(inter, dim, bias=True)


class _VisionBlock(nn.Module):
    def __init__(self, dim, inter):
        super().__init__()
        self.norm1, self.norm2 = _RMSNorm(dim), _RMSNorm(dim)
        self.attn, self.mlp = _VisionAttn(dim), _VisionMLP(dim, inter)


class _Merger(nn.Module):
    def __init__(self, out_dim, ctx_dim, merge):
        super().__init__()
        self.hidden_size = ctx_dim * merge * merge
        self.ln_q = _RMSNorm(ctx_dim)
        self.mlp = nn.Sequential(nn.Linear(self.hidden_size, self.hidden_size), nn.GELU(), nn.Linear(self.hidden_size, out_dim))


class Qwen2_5_VisionTransformerPretrainedModel(nn.Module):
    def __init__(self, config):
        super().__init__()
        g = lambda k, d=None: getattr(config, k, d) if not isinstance(config, dict) else config.get(k, d)   # noqa: E731
        self.hidden, self.depth, self.heads = g("hidden_size"), g("depth"), g("num_heads")
        self.inter, self.out_hidden = g("intermediate_size"), g("out_hidden_size")
        self.patch_size, self.tpatch, self.in_ch = g("patch_size"), g("temporal_patch_size", 2), g("in_channels", None) or g("in_chans", 3)
        self.spatial_merge_size, self.window_size = g("spatial_merge_size", 2), g("window_size", 112)
        self.fullatt_block_indexes = list(g("fullatt_block_indexes", []))
        self.spatial_merge_unit = self.spatial_merge_size ** 2
        self.patch_embed = _PatchEmbed(self.patch_size, self.tpatch, self.in_ch, self.hidden)
        self.blocks = nn.ModuleList([_VisionBlock(self.hidden, self.inter) for _ in range(self.depth)])
        self.merger = _Merger(self.out_hidden, self.hidden, self.spatial_merge_size)
        self._prep = None

    @property
    def dtype(self):
        return self.patch_embed.proj.weight.dtype

    def _prepared(self):
        """GEMM-ready views, rebuilt when a parameter changes: flat patch weight, gate|up stacked, down padded to K % 8 == 0, fp32 biases"""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._prep is not None and self._prep["key"] == key:
            return self._prep
        for p in self.parameters():
            if p.dtype != torch.bfloat16 or not p.is_cuda:
                raise L.UmoeError("the vision tower runs on the HIP path only: parameters must be bf16 on a ROCm device")
        Ip = (self.inter + 7) // 8 * 8
        P = dict(key=key, Ip=Ip, patch=self.patch_embed.proj.weight.data.reshape(self.hidden, -1).contiguous(), blocks=[])
        for b in self.blocks:
            dn = torch.zeros((self.hidden, Ip), dtype=torch.bfloat16, device=b.mlp.down_proj.weight.device)
            dn[:, : self.inter] = b.mlp.down_proj.weight.data
            P["blocks"].append(dict(
                qkv_w=b.attn.qkv.weight.data.contiguous(), qkv_b=b.attn.qkv.bias.data.float().contiguous(),
                proj_w=b.attn.proj.weight.data.contiguous(), proj_b=b.attn.proj.bias.data.float().contiguous(),
                gu_w=torch.cat([b.mlp.gate_proj.weight.data, b.mlp.up_proj.weight.data], 0).contiguous(),
                gu_b=torch.cat([b.mlp.gate_proj.bias.data, b.mlp.up_proj.bias.data], 0).float().contiguous(),
                dn_w=dn, dn_b=b.mlp.down_proj.bias.data.float().contiguous()))
        m0, m2 = self.merger.mlp[0], self.merger.mlp[2]
        P["m0_w"], P["m0_b"] = m0.weight.data.contiguous(), m0.bias.data.float().contiguous()
        P["m2_w"], P["m2_b"] = m2.weight.data.contiguous(), m2.bias.data.float().contiguous()
        self._prep = P
        return P

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, grid_thw: torch.Tensor, **unused) -> torch.Tensor:
        """pixel patches [n_patches, in_ch * temporal_patch * patch^2] -> merged embeddings [n_patches / merge^2, out_hidden]"""
        if not hidden_states.is_cuda:
            raise L.UmoeError("the vision tower needs device tensors (there is no CPU path in the product)")
        dev, lib = hidden_states.device, L.lib()
        P = self._prepared()
        grid = grid_thw.cpu()
        S, unit, H, hd = hidden_states.shape[0], self.spatial_merge_unit, self.heads, self.hidden // self.heads
        # host tables: window order, rotary angles (fp32 table like a model loaded in bf16: the buffer is created in float), segments
        widx, cu_win = window_index(grid, self.window_size, self.spatial_merge_size, self.patch_size)
        pos = rot_pos_ids(grid, self.spatial_merge_size)
        dim = hd // 2
        inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float) / dim))
        table = torch.outer(torch.arange(int(grid[:, 1:].max()), dtype=torch.float), inv)
        rot = table[pos].flatten(1).reshape(S // unit, unit, -1)[widx].reshape(S, -1)
        emb = torch.cat((rot, rot), -1)
        cos, sin = emb.cos().to(dev).contiguous(), emb.sin().to(dev).contiguous()
        cu_full = [0] + torch.repeat_interleave(grid[:, 1] * grid[:, 2], grid[:, 0]).cumsum(0).tolist()

        def segments(cu):
            lo = torch.empty(S, dtype=torch.int32)
            hi = torch.empty(S, dtype=torch.int32)
            for a, b in zip(cu[:-1], cu[1:]):
                lo[a:b], hi[a:b] = a, b
            return lo.to(dev), hi.to(dev)
        seg_win, seg_full = segments(cu_win), segments(cu_full)
        x = ops.tlinear(hidden_states.to(torch.bfloat16).contiguous(), P["patch"])                       # Conv3D == linear on the flat patch
        x = x.reshape(S // unit, unit, -1)[widx.to(dev)].reshape(S, -1).contiguous()
        scale = float(hd) ** -0.5
        for li, (blk, W) in enumerate(zip(self.blocks, P["blocks"])):
            y = ops.rmsnorm(x, blk.norm1.weight.data, blk.norm1.variance_epsilon)
            qkv = ops.tlinear(y, W["qkv_w"], bias=W["qkv_b"])
            L.check(lib.umoe_vision_rope(qkv.data_ptr(), cos.data_ptr(), sin.data_ptr(), S, H, hd, _stream()), "umoe_vision_rope")
            lo, hi = seg_full if li in self.fullatt_block_indexes else seg_win
            ao = torch.empty((S, self.hidden), dtype=torch.bfloat16, device=dev)
            L.check(lib.umoe_vision_attn(qkv.data_ptr(), lo.data_ptr(), hi.data_ptr(), S, H, hd, scale, ao.data_ptr(), _stream()), "umoe_vision_attn")
            x = ops.tlinear(ao, W["proj_w"], bias=W["proj_b"], resid=x)
            y = ops.rmsnorm(x, blk.norm2.weight.data, blk.norm2.variance_epsilon)
            gu = ops.tlinear(y, W["gu_w"], bias=W["gu_b"])
            h = torch.empty((S, P["Ip"]), dtype=torch.bfloat16, device=dev)
            L.check(lib.umoe_swiglu_pair(gu.data_ptr(), S, self.inter, P["Ip"], h.data_ptr(), _stream()), "umoe_swiglu_pair")
            x = ops.tlinear(h, W["dn_w"], bias=W["dn_b"], resid=x)
        z = ops.rmsnorm(x, self.merger.ln_q.weight.data, self.merger.ln_q.variance_epsilon).view(-1, self.hidden * unit)
        z = ops.tlinear(z, P["m0_w"], bias=P["m0_b"])
        L.check(lib.umoe_gelu(z.data_ptr(), z.numel(), _stream()), "umoe_gelu")
        z = ops.tlinear(z, P["m2_w"], bias=P["m2_b"])
        return z[torch.argsort(widx).to(dev)]


def scatter_vision_embeddings(inputs_embeds: torch.Tensor, input_ids: torch.Tensor, token_id: int, embeds: torch.Tensor, what: str):
    """masked_scatter of image / video embeddings at their placeholder tokens (reference model.py:708-751), with the reference's check"""
    mask = input_ids == token_id
    n_tok, n_feat = int(mask.sum()), embeds.shape[0]
    if n_tok != n_feat:
        raise ValueError(f"{what} features and {what.lower()} tokens do not match: tokens: {n_tok}, features {n_feat}")
    return inputs_embeds.masked_scatter(mask.unsqueeze(-1).expand_as(inputs_embeds), embeds.to(inputs_embeds.device, inputs_embeds.dtype))


# ----------------------------------------------------------------------------------------------- frames -> processor patch layout
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def smart_resize(h: int, w: int, factor: int = 28, min_pixels: int = 4 * 28 * 28, max_pixels: int = 64 * 28 * 28) -> Tuple[int, int]:
    """sides rounded to multiples of `factor`, area inside [min_pixels, max_pixels], aspect ratio kept (the Qwen2-VL processor's rule,
    called through qwen_vl_utils at utils/UniMoE_Audio_mod.py:158-195 with the pixel budget of mod.py:49-53)"""
    hb, wb = max(factor, round(h / factor) * factor), max(factor, round(w / factor) * factor)
    if hb * wb > max_pixels:
        beta = math.sqrt((h * w) / max_pixels)
        hb, wb = max(factor, math.floor(h / beta / factor) * factor), max(factor, math.floor(w / beta / factor) * factor)
    elif hb * wb < min_pixels:
        beta = math.sqrt(min_pixels / (h * w))
        hb, wb = math.ceil(h * beta / factor) * factor, math.ceil(w * beta / factor) * factor
    return hb, wb


def frames_to_patches(frames: torch.Tensor, patch: int = 14, tpatch: int = 2, merge: int = 2, max_pixels: int = 64 * 28 * 28,
                      min_pixels: int = 4 * 28 * 28) -> Tuple[torch.Tensor, torch.Tensor]:
    """[F, H, W, 3] or [F, 3, H, W] (uint8 or float in [0, 1]) -> (patches [grid_t * grid_h * grid_w, 3 * tpatch * patch^2] float32,
    grid_thw [3]) in the processor's order: temporal pairs, merge x merge blocks of patches, then (channel, time, row, column) inside a
    patch -- the layout the reference's patch embedding unflattens (utils.py:719-725).  PARITY UNPINNED for the resize filter (the
    third-party processor needs torchvision / PIL, absent offline): bicubic with antialiasing via torch."""
    x = frames
    if x.shape[-1] == 3 and x.shape[1] != 3:
        x = x.permute(0, 3, 1, 2)
    x = x.float() / (255.0 if frames.dtype == torch.uint8 else 1.0)
    F_, _, h, w = x.shape
    hb, wb = smart_resize(h, w, patch * merge, min_pixels, max_pixels)
    if (hb, wb) != (h, w):
        x = torch.nn.functional.interpolate(x, size=(hb, wb), mode="bicubic", align_corners=False, antialias=True).clamp(0, 1)
    mean, std = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    x = (x - mean.to(x.device)) / std.to(x.device)
    if F_ % tpatch:
        x = torch.cat([x, x[-1:].expand(tpatch - F_ % tpatch, -1, -1, -1)], 0)       # the last frame repeats to fill the temporal patch
    gt, gh, gw = x.shape[0] // tpatch, hb // patch, wb // patch
    x = x.reshape(gt, tpatch, 3, gh // merge, merge, patch, gw // merge, merge, patch)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)
    return x.reshape(gt * gh * gw, 3 * tpatch * patch * patch).contiguous(), torch.tensor([gt, gh, gw])
