"""Tensor-level wrappers over the C-ABI (PyTorch-ROCm tensors are containers only).

Every function requires device tensors and launches on torch's current HIP stream.  No function
here computes on the CPU: a missing library or a CPU tensor is an error, never a silent fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _lib as L

PRO_PLAIN, PRO_RMSNORM = 0, 1
EPI_BF16, EPI_BF16_RESID, EPI_SWIGLU, EPI_F32, EPI_F32_RAW = 0, 1, 2, 3, 4


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pv(t: Optional[torch.Tensor]):
    """pointer of a 2-d row-strided view (unit stride in the last dimension); the kernel gets the row stride separately"""
    if t is None:
        return None
    if not t.is_cuda:
        raise L.UmoeError("umoe ops need device tensors (there is no CPU path in the product)")
    if t.dim() != 2 or t.stride(1) != 1:
        raise L.UmoeError("umoe ops need unit stride in the last dimension")
    return t.data_ptr()


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise L.UmoeError("umoe ops need device tensors (there is no CPU path in the product)")
    if not t.is_contiguous():
        raise L.UmoeError("umoe ops need contiguous tensors")
    return C.c_void_p(t.data_ptr())


def _bf(t):
    assert t.dtype == torch.bfloat16, t.dtype
    return t


# ----------------------------------------------------------------------------- packing
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """nn.Linear weight [N, K] bf16 -> WP16 packed (see include/umoe.h)."""
    _bf(w)
    N, K = w.shape
    out = torch.empty(L.lib().umoe_packed_elems(N, K), dtype=torch.bfloat16, device=w.device)
    L.check(L.lib().umoe_pack_weight(_p(w.contiguous()), N, K, _p(out), _stream()), "umoe_pack_weight")
    return out


def pack_gate_up(wg: torch.Tensor, wu: torch.Tensor) -> torch.Tensor:
    _bf(wg), _bf(wu)
    I, K = wg.shape
    assert wu.shape == wg.shape
    out = torch.empty(2 * I * K, dtype=torch.bfloat16, device=wg.device)
    L.check(L.lib().umoe_pack_gate_up(_p(wg.contiguous()), _p(wu.contiguous()), I, K, _p(out), _stream()), "umoe_pack_gate_up")
    return out


# ----------------------------------------------------------------------------- router
def router_fwd(x: Optional[torch.Tensor], gate_w: Optional[torch.Tensor], *, n_dyn: int, n_real: int, n_fix: int,
               top_p: float, fixed_top_k: int = 0, jitter_eps: float = 0.01, logits_in: Optional[torch.Tensor] = None,
               attn_mask: Optional[torch.Tensor] = None, norm_w: Optional[torch.Tensor] = None, rms_eps: float = 1e-6,
               want_h: bool = False, logits_bf16: Optional[bool] = None, gumbel: Optional[torch.Tensor] = None,
               rand_u: Optional[torch.Tensor] = None, x_noise: Optional[torch.Tensor] = None) -> dict:
    """gumbel [S, n_dyn, n_dyn] / rand_u [S, n_dyn] fp32: the training branch of the mixer (core.py:111-137) with its noise as an
    input; the result then carries `round_factor` [S, n_dyn] (mask_for_one per round) for router_bwd.
    x_noise [S, D] fp32: the input jitter in front of the fp32 gate (core.py:240-249) -- the gate sees float(x) * x_noise unrounded."""
    E = n_dyn + n_fix
    if logits_in is not None:
        S = logits_in.shape[0]
        dev = logits_in.device
        is_bf16 = logits_in.dtype == torch.bfloat16
        D = 0
    else:
        S, D = x.shape
        dev = x.device
        is_bf16 = True if logits_bf16 is None else logits_bf16
    o = dict(
        logits=torch.empty((S, E), dtype=torch.bfloat16 if is_bf16 else torch.float32, device=dev),
        top_k=torch.empty((S,), dtype=torch.int64, device=dev),
        sel=torch.empty((S, n_dyn), dtype=torch.int32, device=dev),
        expert_mask=torch.empty((S, E), dtype=torch.int32, device=dev),
        routing_weights=torch.empty((S, n_dyn), dtype=torch.float32, device=dev),
        global_weight=torch.empty((S, E), dtype=torch.float32, device=dev),
        moe_weight=torch.empty((S, n_real), dtype=torch.float32, device=dev),
    )
    h = torch.empty((S, D), dtype=torch.bfloat16, device=dev) if (want_h and x is not None) else None
    am = None
    if attn_mask is not None:
        am = attn_mask.reshape(-1).to(torch.uint8).contiguous()
    a = L.RouterArgs(
        x=_p(x), gate_w=_p(gate_w), norm_w=_p(norm_w), h_out=_p(h), logits_in=_p(logits_in), attn_mask=_p(am), S=S, D=D,
        n_dyn=n_dyn, n_real=n_real, n_fix=n_fix, logits_bf16=int(is_bf16), top_p=top_p, fixed_top_k=int(fixed_top_k),
        jitter_eps=jitter_eps, rms_eps=rms_eps, logits_out=_p(o["logits"]), top_k=_p(o["top_k"]), sel=_p(o["sel"]),
        expert_mask=_p(o["expert_mask"]), routing_w=_p(o["routing_weights"]), global_w=_p(o["global_weight"]),
        moe_w=_p(o["moe_weight"]))
    if x_noise is not None:
        assert x is not None and x_noise.dtype == torch.float32 and tuple(x_noise.shape) == (S, D) and x_noise.is_contiguous()
        a.x_noise = x_noise.data_ptr()
    if gumbel is not None:
        assert rand_u is not None and tuple(gumbel.shape) == (S, n_dyn, n_dyn) and tuple(rand_u.shape) == (S, n_dyn)
        gk, uk = gumbel.float().contiguous(), rand_u.float().contiguous()
        o["round_factor"] = torch.empty((S, n_dyn), dtype=torch.float32, device=dev)
        a.gumbel, a.rand_u, a.round_factor = gk.data_ptr(), uk.data_ptr(), o["round_factor"].data_ptr()
    L.check(L.lib().umoe_router_fwd(C.byref(a), _stream()), "umoe_router_fwd")
    if h is not None:
        o["h"] = h
    return o


def router_dispatch_fwd(x, gate_w, *, n_dyn, n_real, n_fix, top_p, fixed_top_k=0, jitter_eps=0.01, norm_w=None,
                        rms_eps=1e-6) -> dict:
    """router + ragged dispatch tables; one fused launch when S <= 16 (the decode shape)."""
    S, D = x.shape
    E = n_dyn + n_fix
    dev = x.device
    o = dict(logits=torch.empty((S, E), dtype=torch.bfloat16, device=dev), top_k=torch.empty((S,), dtype=torch.int64, device=dev),
             sel=torch.empty((S, n_dyn), dtype=torch.int32, device=dev), expert_mask=torch.empty((S, E), dtype=torch.int32, device=dev),
             routing_weights=torch.empty((S, n_dyn), dtype=torch.float32, device=dev),
             global_weight=torch.empty((S, E), dtype=torch.float32, device=dev),
             moe_weight=torch.empty((S, n_real), dtype=torch.float32, device=dev),
             counts=torch.zeros(16, dtype=torch.int32, device=dev), offsets=torch.zeros(17, dtype=torch.int32, device=dev),
             slot_token=torch.zeros(max(1, S * n_real), dtype=torch.int32, device=dev),
             slot_of=torch.empty((S, n_real), dtype=torch.int32, device=dev), h=torch.empty((S, D), dtype=torch.bfloat16, device=dev))
    a = L.RouterArgs(x=_p(x), gate_w=_p(gate_w), norm_w=_p(norm_w), h_out=_p(o["h"]), logits_in=None, attn_mask=None, S=S, D=D,
                     n_dyn=n_dyn, n_real=n_real, n_fix=n_fix, logits_bf16=1, top_p=top_p, fixed_top_k=int(fixed_top_k),
                     jitter_eps=jitter_eps, rms_eps=rms_eps, logits_out=_p(o["logits"]), top_k=_p(o["top_k"]), sel=_p(o["sel"]),
                     expert_mask=_p(o["expert_mask"]), routing_w=_p(o["routing_weights"]), global_w=_p(o["global_weight"]),
                     moe_w=_p(o["moe_weight"]))
    L.check(L.lib().umoe_router_dispatch_fwd(C.byref(a), _p(o["counts"]), _p(o["offsets"]), _p(o["slot_token"]), _p(o["slot_of"]),
                                             _stream()), "umoe_router_dispatch_fwd")
    return o


def aux_loss(logits: torch.Tensor, expert_mask: torch.Tensor, n_dyn: int, token_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """reference audio_load_balancing_loss_func (core.py:361-389) as one HIP launch; returns a 0-dim fp32 tensor."""
    S, E = logits.shape
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    tw = None if token_weight is None else token_weight.reshape(-1).float().contiguous()
    if S >= 2048:       # many tokens: the two-launch form (64 workgroups + a finisher)
        ws = torch.empty(int(L.lib().umoe_aux_loss_workspace_floats()), dtype=torch.float32, device=logits.device)
        L.check(L.lib().umoe_aux_loss_fwd_ws(_p(logits.contiguous()), int(logits.dtype == torch.bfloat16), _p(expert_mask), _p(tw), S, E, n_dyn,
                                             _p(out), _p(ws), _stream()), "umoe_aux_loss_fwd_ws")
        return out[0]
    L.check(L.lib().umoe_aux_loss_fwd(_p(logits.contiguous()), int(logits.dtype == torch.bfloat16), _p(expert_mask), _p(tw), S, E, n_dyn,
                                      _p(out), _stream()), "umoe_aux_loss_fwd")
    return out[0]


def expert_capacity(num_tokens: int, num_experts: int, capacity_factor: float, min_capacity: int) -> int:
    """reference _audio_expert_capacity (core.py:170-175) in its float32 tensor arithmetic: max(ceil(S / E * cf), min_capacity)."""
    import numpy as np
    c = int(np.ceil(np.float32(num_tokens / num_experts) * np.float32(capacity_factor)))
    return max(c, int(min_capacity))


def token_drop(logits: torch.Tensor, expert_mask: torch.Tensor, routing_w: torch.Tensor, *, n_dyn: int, n_real: int, n_fix: int,
               capacity: int, policy: str) -> dict:
    """Token-drop branch of the block (core.py:302-329) as two HIP launches; returns the post-drop expert_mask, the renormalised
    routing weights and the recomputed global / MoE weights."""
    if policy not in ("probs", "position"):
        raise ValueError(f"Invalid drop_policy: {policy}")                 # core.py:325
    S, E = logits.shape
    dev = logits.device
    o = dict(expert_mask=torch.empty((S, E), dtype=torch.int32, device=dev),
             routing_weights=torch.empty((S, n_dyn), dtype=torch.float32, device=dev),
             global_weight=torch.empty((S, E), dtype=torch.float32, device=dev),
             moe_weight=torch.empty((S, n_real), dtype=torch.float32, device=dev))
    L.check(L.lib().umoe_token_drop(_p(logits.contiguous()), int(logits.dtype == torch.bfloat16), _p(expert_mask), _p(routing_w), S, n_dyn, n_real,
                                    n_fix, int(capacity), 0 if policy == "probs" else 1, _p(o["expert_mask"]), _p(o["routing_weights"]),
                                    _p(o["global_weight"]), _p(o["moe_weight"]), _stream()), "umoe_token_drop")
    return o


def dispatch_build(expert_mask: torch.Tensor, n_real: int) -> dict:
    S, ld = expert_mask.shape
    dev = expert_mask.device
    o = dict(counts=torch.zeros(16, dtype=torch.int32, device=dev), offsets=torch.zeros(17, dtype=torch.int32, device=dev),
             slot_token=torch.zeros(max(1, S * n_real), dtype=torch.int32, device=dev),
             slot_of=torch.empty((S, n_real), dtype=torch.int32, device=dev))
    L.check(L.lib().umoe_dispatch_build(_p(expert_mask), S, ld, n_real, _p(o["counts"]), _p(o["offsets"]),
                                        _p(o["slot_token"]), _p(o["slot_of"]), _stream()), "umoe_dispatch_build")
    return o


def dispatch_build_aligned(expert_mask: torch.Tensor, n_real: int, align: int = 8) -> dict:
    """Dispatch tables with every expert's slot range starting on a multiple of `align` (training)."""
    S, ld = expert_mask.shape
    dev = expert_mask.device
    cap = S * n_real + n_real * (align - 1)
    o = dict(counts=torch.zeros(16, dtype=torch.int32, device=dev), offsets=torch.zeros(17, dtype=torch.int32, device=dev),
             slot_token=torch.zeros(max(1, cap), dtype=torch.int32, device=dev),
             slot_of=torch.empty((S, n_real), dtype=torch.int32, device=dev), cap=cap)
    L.check(L.lib().umoe_dispatch_build_aligned(_p(expert_mask), S, ld, n_real, align, _p(o["counts"]), _p(o["offsets"]),
                                                _p(o["slot_token"]), _p(o["slot_of"]), _stream()), "umoe_dispatch_build_aligned")
    return o


def permute_fwd(x: torch.Tensor, disp: dict, n_real: int) -> torch.Tensor:
    S, D = x.shape
    out = torch.zeros((S * n_real, D), dtype=x.dtype, device=x.device)
    L.check(L.lib().umoe_permute_fwd(_p(x), D, _p(disp["slot_token"]), _p(disp["offsets"][n_real:n_real + 1].contiguous()),
                                     S * n_real, _p(out), _stream()), "umoe_permute_fwd")
    return out


# ----------------------------------------------------------------------------- grouped GEMM
class GroupTable:
    """Device-resident array of umoe_group_t built from Python dicts (kept alive with its tensors)."""

    def __init__(self, groups: Sequence[dict], device):
        arr = (L.Group * len(groups))()
        self.keep = []
        for i, g in enumerate(groups):
            for k in ("w", "bias", "rows", "row_off", "count"):
                t = g.get(k)
                if t is not None:
                    self.keep.append(t)
                    setattr(arr[i], k, t.data_ptr())
            arr[i].static_count = int(g.get("static_count", 0))
            arr[i].a_row_base = int(g.get("a_row_base", 0))
            arr[i].out_row_base = int(g.get("out_row_base", 0))
            arr[i].n_blocks = int(g["n_blocks"])
            arr[i].k = int(g["k"])
            arr[i].a_col_off = int(g.get("a_col_off", 0))
        self.host = arr          # host copy: the descriptors also travel by value in the kernel arguments
        raw = bytes(arr)
        self.dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.n = len(groups)
        self.max_n_blocks = max(int(g["n_blocks"]) for g in groups)
        self.max_k = max(int(g["k"]) for g in groups)


def grouped_gemm(table: GroupTable, a: torch.Tensor, out: torch.Tensor, *, max_rows: int, prologue=PRO_PLAIN,
                 epilogue=EPI_BF16, norm_w=None, rms_eps=1e-6, resid=None, n_valid=None, nt=0, waves=0, ksplit=0, part_stride=0,
                 cache_policy=0):
    args = L.GemmArgs(groups=_p(table.dev), num_groups=table.n, max_rows=max_rows, max_n_blocks=table.max_n_blocks,
                      max_k=table.max_k, a=_p(a), lda=a.stride(0), norm_w=_p(norm_w), rms_eps=rms_eps, resid=_p(resid),
                      out=_p(out), ldo=out.stride(-2), n_valid=out.shape[1] if n_valid is None else n_valid,
                      prologue=prologue, epilogue=epilogue, nt=nt, waves=waves, ksplit=ksplit, part_stride=part_stride,
                      groups_host=C.cast(table.host, C.c_void_p), cache_policy=cache_policy)
    L.check(L.lib().umoe_grouped_gemm(C.byref(args), _stream()), "umoe_grouped_gemm")
    return out


def linear(x: torch.Tensor, w_packed: torch.Tensor, N: int, *, bias: Optional[torch.Tensor] = None, norm_w=None,
           rms_eps=1e-6, resid=None, out_f32=False, nt=0) -> torch.Tensor:
    """y = [rmsnorm](x) @ W^T (+bias) (+resid): one dense group."""
    S, K = x.shape
    tab = GroupTable([dict(w=w_packed, bias=bias, static_count=S, n_blocks=(N + 15) // 16, k=K)], x.device)
    out = torch.empty((S, N), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    epi = EPI_F32 if out_f32 else (EPI_BF16_RESID if resid is not None else EPI_BF16)
    return grouped_gemm(tab, x, out, max_rows=S, prologue=PRO_RMSNORM if norm_w is not None else PRO_PLAIN, epilogue=epi,
                        norm_w=norm_w, rms_eps=rms_eps, resid=resid, n_valid=N, nt=nt)


def combine(y_slots, slot_of, moe_w, y_shared, global_w, resid, n_dyn: int, n_fix: int, norm_w=None, rms_eps=1e-6,
            y_parts=None, shared_row0=-1):
    """y_parts: optional fp32 [n_parts, rows, D] partial slabs of a K-split down GEMM (then y_slots may be None and the
    shared experts' rows start at `shared_row0` of the same slabs)."""
    S, n_real = slot_of.shape
    ref_t = y_slots if y_slots is not None else y_parts
    D = ref_t.shape[-1]
    out = torch.empty((S, D), dtype=torch.bfloat16, device=ref_t.device)
    hn = torch.empty_like(out) if norm_w is not None else None
    a = L.CombineArgs(y_slots=_p(y_slots), slot_of=_p(slot_of), moe_w=_p(moe_w), y_shared=_p(y_shared), global_w=_p(global_w),
                      resid=_p(resid), out=_p(out), S=S, D=D, n_real=n_real, n_dyn=n_dyn, n_fix=n_fix, y_parts=_p(y_parts),
                      n_parts=0 if y_parts is None else y_parts.shape[0], part_stride=0 if y_parts is None else y_parts.stride(0),
                      shared_row0=shared_row0, norm_w=_p(norm_w),
                      norm_out=_p(hn), rms_eps=rms_eps)
    L.check(L.lib().umoe_unpermute_combine_fwd(C.byref(a), _stream()), "umoe_unpermute_combine_fwd")
    return (out, hn) if norm_w is not None else out


# ----------------------------------------------------------------------------- norm / rope / attention
def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float, resid: Optional[torch.Tensor] = None):
    S, D = x.shape
    y = torch.empty_like(x)
    s = torch.empty_like(x) if resid is not None else None
    L.check(L.lib().umoe_rmsnorm_residual_fwd(_p(x), _p(resid), _p(w), eps, S, D, _p(s), _p(y), _stream()), "umoe_rmsnorm")
    return (y, s) if resid is not None else y


_ROPE_TABLES: dict = {}


def rope_tables(max_pos: int, head_dim: int, theta: float, device) -> tuple:
    """cos/sin [>= max_pos][hd/2] bf16, built with the same torch ops as Qwen2_5_VLRotaryEmbedding (fp32 -> bf16).  A row depends on its
    position only, so a table is built once per (length rounded up to 1024, head_dim, theta, device) and kept: building it on the host and
    uploading it took 5 ms of every training step (two idle gaps of 2.6 ms in the kernel trace, profiles/r03n)."""
    n = (int(max_pos) + 1023) & ~1023
    key = (n, int(head_dim), float(theta), str(device))
    hit = _ROPE_TABLES.get(key)
    if hit is None:
        inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
        fr = torch.arange(n, dtype=torch.float)[:, None] * inv[None, :]
        hit = (fr.cos().to(torch.bfloat16).to(device), fr.sin().to(torch.bfloat16).to(device))
        if len(_ROPE_TABLES) >= 8:
            _ROPE_TABLES.pop(next(iter(_ROPE_TABLES)))
        _ROPE_TABLES[key] = hit
    return hit


def qkv_mrope_kvappend(qkv, cos_tab, sin_tab, pos3, kv_pos, T, H, KVH, hd, sections, k_cache, v_cache):
    n_tok = qkv.shape[0]
    q = torch.empty((n_tok, H * hd), dtype=torch.bfloat16, device=qkv.device)
    a = L.RopeArgs(qkv=_p(qkv), cos_tab=_p(cos_tab), sin_tab=_p(sin_tab), pos3=_p(pos3), kv_pos=_p(kv_pos), n_tok=n_tok, T=T,
                   H=H, KVH=KVH, hd=hd, sec0=sections[0], sec1=sections[1], sec2=sections[2], Lmax=k_cache.shape[2],
                   q_out=_p(q), k_cache=_p(k_cache), v_cache=_p(v_cache))
    L.check(L.lib().umoe_qkv_mrope_kvappend(C.byref(a), _stream()), "umoe_qkv_mrope_kvappend")
    return q


def attention(q, k_cache, v_cache, kv_start, q_pos0, nq, H, splits=1, qkv_raw=None, cos_tab=None, sin_tab=None, pos3=None,
              sections=(16, 24, 24), lse_out=None):
    """q: rotated queries [rows*nq, H*hd]; or pass qkv_raw (decode, nq == 1) to fuse mRoPE + KV append into the kernel."""
    rows, KVH, Lmax, hd = k_cache.shape
    n = (q if q is not None else qkv_raw).shape[0]
    dev0 = (q if q is not None else qkv_raw).device
    po = torch.empty((n, H, splits, hd), dtype=torch.float32, device=dev0)
    pm = torch.empty((n, H, splits, 2), dtype=torch.float32, device=dev0)
    out = torch.empty((n, H * hd), dtype=torch.bfloat16, device=dev0)
    a = L.AttnArgs(q=_p(q), k_cache=_p(k_cache), v_cache=_p(v_cache), kv_start=_p(kv_start), q_pos0=_p(q_pos0), rows=rows,
                   nq=nq, H=H, KVH=KVH, hd=hd, Lmax=Lmax, splits=splits, scale=float(hd) ** -0.5, part_o=_p(po),
                   part_ml=_p(pm), out=_p(out), qkv_raw=_p(qkv_raw), cos_tab=_p(cos_tab), sin_tab=_p(sin_tab), pos3=_p(pos3),
                   sec0=sections[0], sec1=sections[1], sec2=sections[2], lse_out=_p(lse_out))
    if nq >= 16 and qkv_raw is None:       # many queries per row: MFMA flash-attention kernel (umoe_attn_prefill_fwd)
        L.check(L.lib().umoe_attn_prefill_fwd(C.byref(a), _stream()), "umoe_attn_prefill_fwd")
    else:
        L.check(L.lib().umoe_attn_decode(C.byref(a), _stream()), "umoe_attn_decode")
    return out


# ----------------------------------------------------------------------------- codec side
def codec_embed_sum(tok: torch.Tensor, emb: torch.Tensor) -> torch.Tensor:
    rows, Cc = tok.shape
    _, V, D = emb.shape
    out = torch.empty((rows, D), dtype=torch.bfloat16, device=emb.device)
    L.check(L.lib().umoe_codec_embed_sum(_p(tok.to(torch.int32).contiguous()), _p(emb), rows, Cc, V, D, _p(out), _stream()),
            "umoe_codec_embed_sum")
    return out


def mul_noise(x: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """bf16(x.float() * noise) in one pass (training-time input jitter, core.py:240-244)."""
    x = x.contiguous()
    noise = noise.contiguous()
    assert x.dtype == torch.bfloat16 and noise.dtype == torch.float32 and x.shape == noise.shape
    y = torch.empty_like(x)
    L.check(L.lib().umoe_mul_noise(_p(x), _p(noise), x.numel(), _p(y), _stream()), "umoe_mul_noise")
    return y


def codec_embed_sum_bwd(tok: torch.Tensor, d_out: torch.Tensor, V: int) -> torch.Tensor:
    """[C][V][D] gradient of the stacked codec embedding tables (deterministic: ascending row order, fp32 accumulation)."""
    rows, Cc = tok.shape
    D = d_out.shape[1]
    d_emb = torch.empty((Cc, V, D), dtype=torch.bfloat16, device=d_out.device)
    L.check(L.lib().umoe_codec_embed_sum_bwd(_p(tok.to(torch.int32).contiguous()), _p(d_out.to(torch.bfloat16).contiguous()), rows, Cc, V, D,
                                             _p(d_emb), _stream()), "umoe_codec_embed_sum_bwd")
    return d_emb


def cfg_sample(logits: torch.Tensor, B: int, Cc: int, V: int, *, cfg_scale, temperature, top_p, top_k, eos, eos_mul,
               enable_eos=True, do_sample=True, seed=0, want_probs=False):
    pred = torch.empty((B, Cc), dtype=torch.int64, device=logits.device)
    probs = torch.empty((B * Cc, V), dtype=torch.float32, device=logits.device) if want_probs else None
    a = L.SampleArgs(logits=_p(logits), B=B, C=Cc, V=V, cfg_scale=cfg_scale, temperature=temperature, top_p=top_p,
                     eos_mul=eos_mul, top_k=-1 if top_k is None else top_k, eos=eos, enable_eos=int(enable_eos), min_tokens=-1,
                     step=None, do_sample=int(do_sample), seed=seed, pred=_p(pred), probs_out=_p(probs))
    L.check(L.lib().umoe_codec_head_cfg_sample(C.byref(a), _stream()), "umoe_codec_head_cfg_sample")
    return (pred, probs) if want_probs else pred


def delay_step(pred, tokens, state, delay, eos, pad):
    B, Tmax, Cc = tokens.shape
    L.check(L.lib().umoe_delay_step(_p(pred), _p(tokens), _p(state), _p(delay), B, Cc, Tmax, eos, pad,
                                    int(delay.max().item()), _stream()), "umoe_delay_step")


def rvq_from_codes(codes, codebooks, out_w, out_b):
    NQ, T = codes.shape
    _, CB, cd = codebooks.shape
    Dl = out_w.shape[1]
    z = torch.empty((Dl, T), dtype=torch.float32, device=codes.device)
    L.check(L.lib().umoe_rvq_from_codes(_p(codes.to(torch.int32).contiguous()), _p(codebooks), _p(out_w), _p(out_b), NQ, CB,
                                        cd, Dl, T, _p(z), _stream()), "umoe_rvq_from_codes")
    return z


def rvq_nearest(z, codebooks, in_w, in_b, out_w, out_b):
    Dl, T = z.shape
    NQ, CB, cd = codebooks.shape
    codes = torch.empty((NQ, T), dtype=torch.int32, device=z.device)
    ws = torch.empty((T, Dl), dtype=torch.float32, device=z.device)
    L.check(L.lib().umoe_rvq_nearest(_p(z), _p(codebooks), _p(in_w), _p(in_b), _p(out_w), _p(out_b), NQ, CB, cd, Dl, T,
                                     _p(codes), _p(ws), _stream()), "umoe_rvq_nearest")
    return codes


def codec_ce(logits: torch.Tensor, labels: torch.Tensor, want_grad: bool = False):
    """logits [N, C, V] fp32 (already shifted), labels [N, C] int64 (-100 = ignore) -> (total, ch_loss [C], ch_count [C], dlogits?)
    reference training loss, UniMoE_Audio_model.py:830-847."""
    N, Cc, V = logits.shape
    dev = logits.device
    nll = torch.empty((N, Cc), dtype=torch.float32, device=dev)
    probs = torch.empty((N, Cc, V), dtype=torch.float32, device=dev) if want_grad else None
    ch_loss = torch.empty(Cc, dtype=torch.float32, device=dev)
    ch_cnt = torch.empty(Cc, dtype=torch.int32, device=dev)
    total = torch.empty(1, dtype=torch.float32, device=dev)
    lab = labels.to(torch.int64).contiguous()
    L.check(L.lib().umoe_codec_ce_fwd(_p(logits.contiguous()), _p(lab), N, Cc, V, _p(nll), _p(probs), _p(ch_loss), _p(ch_cnt),
                                      _p(total), _stream()), "umoe_codec_ce_fwd")
    if not want_grad:
        return total[0], ch_loss, ch_cnt
    dl = torch.empty_like(probs)
    L.check(L.lib().umoe_codec_ce_bwd(_p(probs), _p(lab), _p(ch_cnt), N, Cc, V, 1.0, _p(dl), _stream()), "umoe_codec_ce_bwd")
    return total[0], ch_loss, ch_cnt, dl


def prefetch(t: torch.Tensor, wgs: int = 256, nbytes: Optional[int] = None):
    """Warm the Infinity Cache with the bytes of `t` (plain loads, nothing stored)."""
    n = t.numel() * t.element_size() if nbytes is None else nbytes
    L.check(L.lib().umoe_prefetch(_p(t), n, wgs, _stream()), "umoe_prefetch")


def tiled_gemm(groups: Sequence[dict], a: torch.Tensor, out: torch.Tensor, *, max_rows: int, epilogue=EPI_BF16, resid=None,
               aux_out: Optional[torch.Tensor] = None):
    """Compute-bound grouped GEMM on ROW-MAJOR weights (umoe_tiled_gemm): each group dict has w [N, K] (and w2 for SwiGLU),
    optional bias / rows / row_off / count tensors and static_count / a_row_base / out_row_base ints."""
    arr = (L.TGroup * len(groups))()
    keep = []
    for i, g in enumerate(groups):
        w = g["w"]
        assert w.dim() == 2 and w.stride(1) == 1
        for k in ("w", "w2", "bias", "rows", "row_off", "count", "k_off", "k_count"):
            t = g.get(k)
            if t is not None:
                keep.append(t)
                setattr(arr[i], k, t.data_ptr())
        arr[i].static_count = int(g.get("static_count", 0))
        arr[i].a_row_base = int(g.get("a_row_base", 0))
        arr[i].out_row_base = int(g.get("out_row_base", 0))
        if g.get("w_kmajor"):                 # w [K][N]: the contraction index is the weight's row (input gradients on nn.Linear weights as stored)
            arr[i].n, arr[i].k, arr[i].ldw, arr[i].w_kmajor = int(w.shape[1]), int(g.get("k", w.shape[0])), int(w.stride(0)), 1
            arr[i].k_w1 = int(g.get("k_w1", 0))
        else:
            arr[i].n, arr[i].k, arr[i].ldw = int(w.shape[0]), int(g.get("k", w.shape[1])), int(w.stride(0))
        arr[i].a_col_off = int(g.get("a_col_off", 0))
        arr[i].out_col_off = int(g.get("out_col_off", 0))
    args = L.TGemmArgs(groups=C.cast(arr, C.c_void_p), num_groups=len(groups), max_rows=max_rows, a=_pv(a), lda=a.stride(0),
                       resid=_p(resid), out=_pv(out), ldo=out.stride(-2), epilogue=epilogue, aux_out=_p(aux_out),
                       ld_aux=0 if aux_out is None else aux_out.stride(0))
    L.check(L.lib().umoe_tiled_gemm(C.byref(args), _stream()), "umoe_tiled_gemm")
    return out


def tiled_gemm_tn(groups: Sequence[dict], p: torch.Tensor, q: torch.Tensor, out: torch.Tensor, *, k_split: int = 1):
    """out_g[m][n] = sum_k p[k][p_col_off + m] * q[k][q_col_off + n] over the group's row window (umoe_tiled_gemm_tn): the weight-gradient
    product dW = dY^T X on row-major activations, no transposed copies.  Group dict: m, n, optional p / q / out tensors of its own,
    p_col_off, q_col_off, k_off + k (static window) or k_off_dev + k_count_dev (int32 device scalars), out_row_base, out_col_off."""
    arr = (L.TnGroup * len(groups))()
    keep = []
    for i, g in enumerate(groups):
        for k in ("p", "q"):
            t = g.get(k)
            if t is not None:
                keep.append(t)
                setattr(arr[i], k, t.data_ptr())
                setattr(arr[i], "ld" + k, t.stride(0))
        for k in ("out", "k_off_dev", "k_count_dev"):
            t = g.get(k)
            if t is not None:
                keep.append(t)
                setattr(arr[i], k, t.data_ptr())
        for k in ("p_col_off", "q_col_off", "m", "n", "k_off", "k", "out_row_base", "out_col_off"):
            setattr(arr[i], k, int(g.get(k, 0)))
    args = L.TGemmTnArgs(groups=C.cast(arr, C.c_void_p), num_groups=len(groups), p=_pv(p), ldp=p.stride(0), q=_pv(q), ldq=q.stride(0),
                         out=_pv(out), ldo=out.stride(-2), k_split=k_split)
    ws = None
    if k_split > 1 or k_split < 0:                            # (< 0: the library chooses from the tile count and K)
        args.part_stride = out.numel()
        ws = torch.empty(L.lib().umoe_tiled_gemm_tn_workspace_bytes(C.byref(args)), dtype=torch.uint8, device=out.device)
        args.ws = ws.data_ptr()
    L.check(L.lib().umoe_tiled_gemm_tn(C.byref(args), _stream()), "umoe_tiled_gemm_tn")
    return out


def linear_input_grad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dX [S][K] = dy[:, :N] W for y = x W^T with W [N][K] as stored (its row is the contraction index: umoe_tgroup_t.w_kmajor) -- no
    transposed weight copy; dy may carry ZERO-padded columns behind N (row stride a multiple of 8)."""
    S, (N, K) = dy.shape[0], w.shape
    assert dy.shape[1] >= N and dy.stride(0) >= _r8(N)
    out = torch.empty((S, K), dtype=torch.bfloat16, device=dy.device)
    return tiled_gemm([dict(w=w, w_kmajor=1, static_count=S)], dy, out, max_rows=S)


def linear_weight_grad(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """dW [N][K] = dy[:, :N]^T x for y = x W^T (torch.nn.functional.linear backward, core.py:21-49): no transposed copies; dy may carry
    zero-padded columns behind N (row stride a multiple of 8)."""
    N, K = w.shape
    assert x.shape[0] == dy.shape[0] and x.shape[1] == K and dy.shape[1] >= N
    dw = torch.empty((N, K), dtype=torch.bfloat16, device=w.device)
    return tiled_gemm_tn([dict(m=N, n=K, k=x.shape[0])], dy, x, dw, k_split=-1)


_SIDE_STREAMS: dict = {}


def bwd_overlap() -> bool:
    """Independent kernels of one backward call run beside each other on side streams (forked and joined inside the call): UMOE_BWD_OVERLAP,
    default on; the C library reads the same variable for its composites."""
    import os
    return os.environ.get("UMOE_BWD_OVERLAP", "1") != "0"


def side_stream(device, tag: str = "wgrad"):
    key = (str(device), tag)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def linear_grads(dy: torch.Tensor, x: torch.Tensor, w: torch.Tensor, need_dx: bool = True, need_dw: bool = True):
    """(dX, dW) of y = x W^T.  Nothing in a backward pass reads dW, so when both are wanted the weight gradient runs on a SIDE stream beside
    the input gradient (fork / join inside this call, like the expert composite in umoe_bwd.hip): the round tails of the two GEMMs and the
    K-split reduction overlap.  UMOE_BWD_OVERLAP=0: both on the current stream."""
    if not (need_dx and need_dw) or not bwd_overlap():
        return (linear_input_grad(dy, w) if need_dx else None), (linear_weight_grad(dy, x, w) if need_dw else None)
    main = torch.cuda.current_stream()
    side = side_stream(dy.device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dw = linear_weight_grad(dy, x, w)
    dy.record_stream(side)          # (the caching allocator must not hand these blocks out again before the side stream is through)
    x.record_stream(side)
    dx = linear_input_grad(dy, w)
    main.wait_stream(side)
    dw.record_stream(main)
    return dx, dw


def tlinear(x: torch.Tensor, w: torch.Tensor, *, bias: Optional[torch.Tensor] = None, resid=None, out_f32=False) -> torch.Tensor:
    """y = x @ w^T (+bias) (+resid) with row-major w [N, K]: the tiled MFMA path for many rows."""
    S = x.shape[0]
    out = torch.empty((S, w.shape[0]), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    epi = EPI_F32 if out_f32 else (EPI_BF16_RESID if resid is not None else EPI_BF16)
    return tiled_gemm([dict(w=w, bias=bias, static_count=S)], x, out, max_rows=S, epilogue=epi, resid=resid)


# ----------------------------------------------------------------------------- backward pieces (training)
def _r8(n: int) -> int:
    return (n + 7) & ~7


def transpose_slots(src: torch.Tensor, dst: torch.Tensor, *, rows=None, counts=None, offsets=None, n_groups: int = 1,
                    max_rows: Optional[int] = None, C_cols: Optional[int] = None):
    """dst[c][off_g + r] = src[row(off_g + r)][c] (zeros up to the next multiple of 8 columns); plain transpose of the
    first max_rows rows when counts/offsets are None.  dst may be a column view of a wider buffer (stride(0) % 8 == 0)."""
    ncol = src.shape[1] if C_cols is None else C_cols
    mr = src.shape[0] if max_rows is None else max_rows
    L.check(L.lib().umoe_transpose_slots(_pv(src), src.stride(0), ncol, _p(rows), _p(counts), _p(offsets), n_groups, mr, _pv(dst),
                                         dst.stride(0), _stream()), "umoe_transpose_slots")
    return dst


def transpose(src: torch.Tensor) -> torch.Tensor:
    """[R][C] -> [C][roundup8(R)] (zero padded columns): transposed weight copies for the input-gradient GEMMs."""
    R, Cc = src.shape
    dst = torch.empty((Cc, _r8(R)), dtype=src.dtype, device=src.device)
    return transpose_slots(src, dst)


def swiglu_bwd(dh: torch.Tensor, gu: torch.Tensor, I: int, dgu: torch.Tensor, *, total_rows: Optional[torch.Tensor], max_rows: int):
    L.check(L.lib().umoe_swiglu_bwd(_p(dh), dh.stride(0), _p(gu), gu.stride(0), I, _p(total_rows), max_rows, _p(dgu), dgu.stride(0),
                                    _stream()), "umoe_swiglu_bwd")
    return dgu


def combine_bwd(dout, y_slots, slot_of, moe_w, y_shared, global_w, n_dyn: int, n_fix: int, dy_slots, dy_shared):
    S, n_real = slot_of.shape
    D = dout.shape[-1]
    d_mw = torch.empty((S, n_real), dtype=torch.float32, device=dout.device)
    d_gs = torch.empty((S, max(1, n_fix)), dtype=torch.float32, device=dout.device) if n_fix else None
    a = L.CombineArgs(y_slots=_p(y_slots), slot_of=_p(slot_of), moe_w=_p(moe_w), y_shared=_p(y_shared), global_w=_p(global_w),
                      S=S, D=D, n_real=n_real, n_dyn=n_dyn, n_fix=n_fix, shared_row0=-1)
    L.check(L.lib().umoe_unpermute_combine_bwd(_p(dout), C.byref(a), _p(dy_slots), _p(dy_shared), _p(d_mw), _p(d_gs), _stream()),
            "umoe_unpermute_combine_bwd")
    return d_mw, d_gs


def permute_bwd(dxe, slot_of, dx_shared, n_fix: int, extra=None):
    S, n_real = slot_of.shape
    D = dxe.shape[-1]
    dx = torch.empty((S, D), dtype=torch.bfloat16, device=dxe.device)
    L.check(L.lib().umoe_permute_bwd(_p(dxe), _p(slot_of), n_real, _p(dx_shared), n_fix, S, D, _p(extra), _p(dx), _stream()),
            "umoe_permute_bwd")
    return dx


def router_bwd(logits, sel, top_k, expert_mask, d_moe_w, d_gw_shared, d_logits_in, n_dyn: int, n_real: int, n_fix: int,
               jitter_eps: float, token_drop: bool = False, round_factor: Optional[torch.Tensor] = None) -> torch.Tensor:
    """token_drop: `expert_mask` is the mask AFTER the drop and the extra renormalisation of core.py:328-329 is in the graph.
    round_factor: mask_for_one per (token, round) of the differentiable-router forward (AudioMoERoutingFunction.backward)."""
    S, E = logits.shape
    out = torch.empty((S, E), dtype=torch.float32, device=logits.device)
    L.check(L.lib().umoe_router_bwd_ex(_p(logits), int(logits.dtype == torch.bfloat16), _p(sel), _p(top_k), _p(expert_mask), _p(d_moe_w),
                                       _p(d_gw_shared), _p(d_logits_in), S, n_dyn, n_real, n_fix, float(jitter_eps), int(bool(token_drop)),
                                       _p(round_factor), _p(out), _stream()), "umoe_router_bwd_ex")
    return out


def aux_loss_bwd(logits, expert_mask, n_dyn: int, token_weight, d_aux: torch.Tensor) -> torch.Tensor:
    S, E = logits.shape
    out = torch.empty((S, E), dtype=torch.float32, device=logits.device)
    ws = torch.empty(17, dtype=torch.float32, device=logits.device)
    tw = None if token_weight is None else token_weight.reshape(-1).float().contiguous()
    L.check(L.lib().umoe_aux_loss_bwd(_p(logits), int(logits.dtype == torch.bfloat16), _p(expert_mask), _p(tw), S, E, n_dyn,
                                      _p(d_aux.reshape(1).float().contiguous()), _p(out), _p(ws), _stream()), "umoe_aux_loss_bwd")
    return out


def rmsnorm_bwd(h: torch.Tensor, w: torch.Tensor, dy: torch.Tensor, eps: float, dsum: Optional[torch.Tensor] = None):
    """Qwen2RMSNorm backward: returns (dh, dw)."""
    S, D = h.shape
    dh = torch.empty_like(h)
    dw = torch.empty_like(w)
    ws = torch.empty(min(S, 512) * D, dtype=torch.float32, device=h.device)
    L.check(L.lib().umoe_rmsnorm_residual_bwd(_p(h), _p(w), _p(dy), _p(dsum), eps, S, D, _p(dh), _p(dw), _p(ws), ws.numel(), _stream()),
            "umoe_rmsnorm_residual_bwd")
    return dh, dw


def experts_swiglu_bwd(ws_list, *, x, h, gu, dy, dx_slots, D: int, I: int, max_rows: int, counts=None, offsets=None, slot_token=None,
                       row_base: int = 0):
    """umoe_grouped_swiglu_bwd (counts/offsets given: routed experts) or umoe_shared_swiglu_bwd (static groups).
    ws_list: list of (w_gate [I,D], w_up [I,D], w_down [D,I]) row-major bf16 tensors.  Returns lists (dWg, dWu, dWd)."""
    G = len(ws_list)
    dev = x.device
    arr = lambda ts: (C.c_void_p * G)(*[t.data_ptr() for t in ts])
    # one stacked allocation per kind: the composite then runs each kind's G weight-gradient products as ONE grouped launch
    # (gate and up in ONE allocation [2][G][I][D]: the shared experts' K-split weight-gradient launch writes them as one slab)
    dwgu = torch.empty((2, G) + tuple(ws_list[0][0].shape), dtype=torch.bfloat16, device=dev)
    dwg, dwu = list(dwgu[0].unbind(0)), list(dwgu[1].unbind(0))
    dwd = list(torch.empty((G,) + tuple(ws_list[0][2].shape), dtype=torch.bfloat16, device=dev).unbind(0))
    keep = (arr([w[0] for w in ws_list]), arr([w[1] for w in ws_list]), arr([w[2] for w in ws_list]), arr(dwg), arr(dwu), arr(dwd))
    a = L.SwigluBwdArgs(num_groups=G, w_gate=keep[0], w_up=keep[1], w_down=keep[2], D=D, I=I, counts=_p(counts), offsets=_p(offsets),
                        slot_token=_p(slot_token), max_rows=max_rows, slot_rows=h.shape[0], row_base=row_base,
                        x=_p(x), ldx=x.stride(0), h=_pv(h), ldh=h.stride(0), gu=_pv(gu), ldgu=gu.stride(0), dy=_pv(dy), lddy=dy.stride(0),
                        dx_slots=_pv(dx_slots), lddx=dx_slots.stride(0), dw_gate=keep[3], dw_up=keep[4], dw_down=keep[5])
    nbytes = L.lib().umoe_swiglu_bwd_workspace_bytes(C.byref(a))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    a.ws, a.ws_bytes = ws.data_ptr(), nbytes
    fn = L.lib().umoe_grouped_swiglu_bwd if counts is not None else L.lib().umoe_shared_swiglu_bwd
    L.check(fn(C.byref(a), _stream()), "umoe_swiglu_bwd (composite)")
    return dwg, dwu, dwd
