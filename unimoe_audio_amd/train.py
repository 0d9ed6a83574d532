"""Training path (BASELINE configs[2]: forward + backward of the 36-layer DCMoE text model + codec head + CE).

torch.autograd.Function wrappers whose forward AND backward run on the HIP kernels behind the C-ABI (include/umoe.h);
torch itself only owns the tape, the tensors and a few glue elementwise ops (residual adds, embedding gathers).
Reference graph: Qwen2_5_VLMoEDecoderLayer.forward utils/UniMoE_Audio_model.py:210-256, text model :319-457,
training loss :817-854; attention / RMSNorm / mRoPE arithmetic = the transformers classes imported at :52-56.

Backward of the contractions: dX = dY W reads the weight as stored (k-major operand of umoe_tiled_gemm, transposing LDS reads) and
dW = dY^T X runs on the row-major activations themselves (umoe_tiled_gemm_tn, transposing LDS reads; K split chosen by the library).  Attention backward: the fused
flash-style kernels of umoe_attn_bwd.hip (the unfused composite of umoe_bwd.hip for shapes they do not cover).  No CPU fallback: CPU
tensors raise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from . import ops


def _pad8(t: torch.Tensor) -> torch.Tensor:
    """rows of a K-contiguous GEMM operand must start on 16-byte boundaries: pad the columns with zeros to a multiple of 8"""
    n = t.shape[1]
    if n % 8 == 0:
        return t
    out = torch.zeros((t.shape[0], ops._r8(n)), dtype=t.dtype, device=t.device)
    out[:, :n] = t
    return out


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ b): nn.Linear on the tiled MFMA GEMM, row-major weight [N][K]."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        return ops.tlinear(x, w, bias=None if b is None else b.float().contiguous())

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = dy.to(torch.bfloat16).contiguous()
        S = x.shape[0]
        dx, dw = ops.linear_grads(_pad8(dy), x, w, ctx.needs_input_grad[0], ctx.needs_input_grad[1])   # [S][N] x [N][K] (weight as stored) | dy^T x
        db = dy.float().sum(0).to(torch.bfloat16) if ctx.has_b and ctx.needs_input_grad[2] else None
        return dx, dw, db


class EmbedFn(torch.autograd.Function):
    """nn.Embedding gather (model.py:655: `embed_tokens(input_ids)`) with the table gradient by the same HIP kernel as the codec tables
    (one channel): torch's sort-based index backward took 3.5 ms per step on the 151 936-row table."""

    @staticmethod
    def forward(ctx, ids, table):
        # ids < 0: rows the caller overwrites afterwards (codec placeholders): they read row 0 and contribute no gradient -- without this
        # the ~6 000 placeholder rows of a training batch all belong to ONE id, whose owner workgroup would add 25 MB of zeros alone
        ctx.save_for_backward(ids)
        ctx.V = table.shape[0]
        return table[ids.clamp(min=0)]

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        D = dy.shape[-1]
        d_tab = ops.codec_embed_sum_bwd(ids.reshape(-1, 1), dy.reshape(-1, D), ctx.V)
        return None, d_tab[0]


class CodecEmbedFn(torch.autograd.Function):
    """sum_c Emb_c[tok[..., c]] (model.py:655-661: bf16 adds in channel order) with the table gradients by one HIP kernel."""

    @staticmethod
    def forward(ctx, tok, *tables):
        emb = torch.stack(tables, 0)                       # [C][V][D]
        ctx.save_for_backward(tok)
        ctx.V = emb.shape[1]
        return ops.codec_embed_sum(tok, emb)

    @staticmethod
    def backward(ctx, dy):
        (tok,) = ctx.saved_tensors
        d_emb = ops.codec_embed_sum_bwd(tok, dy, ctx.V)
        return (None, *d_emb.unbind(0))


class RMSNormFn(torch.autograd.Function):
    """Qwen2RMSNorm: w * bf16(x * rsqrt(mean(x^2) + eps))."""

    @staticmethod
    def forward(ctx, x, w, eps):
        ctx.save_for_backward(x, w)
        ctx.eps = eps
        return ops.rmsnorm(x, w, eps)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw = ops.rmsnorm_bwd(x, w, dy.to(torch.bfloat16).contiguous(), ctx.eps)
        return dx, dw, None


class RopeAttentionFn(torch.autograd.Function):
    """mRoPE + causal GQA attention over one full sequence per row (no cache kept)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, pos3, kv_pos, first_valid, first_valid_host, B, T, H, KVH, hd, sections):
        dev = qkv.device
        kc = torch.empty((B, KVH, T, hd), dtype=torch.bfloat16, device=dev)
        vc = torch.empty_like(kc)
        q = ops.qkv_mrope_kvappend(qkv, cos, sin, pos3, kv_pos, T, H, KVH, hd, sections, kc, vc)
        q0 = torch.zeros(B, dtype=torch.int32, device=dev)
        # log-sum-exp per (query, head): kept for the fused backward (written by the MFMA kernel only: T >= 16, hd 128, G <= 8)
        fused = T >= 16 and hd == 128 and H // KVH <= 8
        lse = torch.empty((B * T, H), dtype=torch.float32, device=dev) if fused else None
        ao = ops.attention(q, kc, vc, first_valid, q0, T, H, splits=1, lse_out=lse)
        ctx.fused = fused
        ctx.save_for_backward(q, kc, vc, cos, sin, pos3, kv_pos, *((ao, lse) if fused else ()))
        ctx.meta = (B, T, H, KVH, hd, tuple(sections), list(first_valid_host))
        return ao

    @staticmethod
    def backward(ctx, d_ao):
        q, kc, vc, cos, sin, pos3, kv_pos, *extra = ctx.saved_tensors
        B, T, H, KVH, hd, sections, fv = ctx.meta
        dev, bf = q.device, torch.bfloat16
        d_ao = d_ao.to(bf).contiguous()
        scale = float(hd) ** -0.5
        lib = L.lib()
        dq = torch.empty_like(q)
        dk = torch.zeros_like(kc)
        dv = torch.zeros_like(vc)
        kv_host = (C.c_int32 * B)(*[int(v) for v in fv])
        a = L.AttnBwdArgs(q=q.data_ptr(), k_cache=kc.data_ptr(), v_cache=vc.data_ptr(), kv_start_host=C.cast(kv_host, C.c_void_p),
                          d_out=d_ao.data_ptr(), rows=B, T=T, H=H, KVH=KVH, hd=hd, Lmax=T, scale=scale, dq=dq.data_ptr(),
                          dk_cache=dk.data_ptr(), dv_cache=dv.data_ptr())
        if ctx.fused:                                   # flash-style backward: needs the forward's output and log-sum-exp
            a.out, a.lse = extra[0].data_ptr(), extra[1].data_ptr()
        nbytes = lib.umoe_attn_prefill_bwd_workspace_bytes(C.byref(a))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        a.ws, a.ws_bytes = ws.data_ptr(), nbytes
        L.check(lib.umoe_attn_prefill_bwd(C.byref(a), ops._stream()), "umoe_attn_prefill_bwd")
        # mRoPE backward + scatter into the QKV row layout
        n_tok = B * T
        dqkv = torch.empty((n_tok, (H + 2 * KVH) * hd), dtype=bf, device=dev)
        a = L.RopeArgs(qkv=None, cos_tab=cos.data_ptr(), sin_tab=sin.data_ptr(), pos3=pos3.data_ptr(), kv_pos=kv_pos.data_ptr(),
                       n_tok=n_tok, T=T, H=H, KVH=KVH, hd=hd, sec0=sections[0], sec1=sections[1], sec2=sections[2], Lmax=T,
                       q_out=None, k_cache=None, v_cache=None)
        L.check(lib.umoe_qkv_mrope_bwd(C.byref(a), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), dqkv.data_ptr(), ops._stream()),
                "umoe_qkv_mrope_bwd")
        return (dqkv,) + (None,) * 12


class CodecCEFn(torch.autograd.Function):
    """sum of the per-channel shifted cross-entropies (model.py:830-847)."""

    @staticmethod
    def forward(ctx, logits, labels):
        total, ch_loss, ch_cnt, dl = ops.codec_ce(logits, labels, want_grad=True)
        ctx.save_for_backward(dl)
        return total

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None


def text_forward_train(model, inputs_embeds: torch.Tensor, attention_mask: Optional[torch.Tensor], position_ids=None,
                       padding_token_mask=None, aux_balance_weight=None):
    """Differentiable counterpart of UniAudioRVQQwen2_5VLMoEForConditionalGeneration.text_forward: returns
    (last_hidden_state [B,T,D], list of per-layer aux losses, per-layer (top_k, expert_mask))."""
    cfg, dev = model.config, inputs_embeds.device
    B, T, D = inputs_embeds.shape
    H, KVH, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    am = torch.ones(B, T, dtype=torch.long, device=dev) if attention_mask is None else attention_mask.to(dev).long()
    if position_ids is None:
        pos = (am.cumsum(-1) - 1).masked_fill(am == 0, 1)
        position_ids = pos[None].expand(3, -1, -1)
    elif position_ids.dim() == 2:
        position_ids = position_ids[None].expand(3, -1, -1)
    pos3 = position_ids.reshape(3, B * T).to(torch.int32).contiguous()
    kv_pos = torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
    first_valid = (am != 0).float().argmax(-1).to(torch.int32).contiguous()
    fv_host = first_valid.tolist()
    cos, sin = ops.rope_tables(int(position_ids.max()) + 2, hd, cfg.rope_theta, dev)
    x = inputs_embeds.reshape(B * T, D).contiguous()
    auxes, masks = [], []
    for layer in model.language_model.layers:
        a = layer.self_attn
        h = RMSNormFn.apply(x, layer.input_layernorm.weight, cfg.rms_norm_eps)
        qkv_w = torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0)
        qkv_b = torch.cat([a.q_proj.bias, a.k_proj.bias, a.v_proj.bias], 0)
        qkv = LinearFn.apply(h, qkv_w, qkv_b)
        ao = RopeAttentionFn.apply(qkv, cos, sin, pos3, kv_pos, first_valid, fv_host, B, T, H, KVH, hd, tuple(cfg.mrope_section))
        x1 = x + LinearFn.apply(ao, a.o_proj.weight, None)                                # model.py:238
        h2 = RMSNormFn.apply(x1, layer.post_attention_layernorm.weight, cfg.rms_norm_eps)
        out = layer.mlp(h2.view(B, T, D), padding_token_mask, aux_balance_weight)          # model.py:241
        x = x1 + out[0].reshape(B * T, D)                                                 # model.py:242
        auxes.append(out[5])
        masks.append((out[2], out[3]))
    hN = RMSNormFn.apply(x, model.language_model.norm.weight, cfg.rms_norm_eps)
    return hN.view(B, T, D), auxes, masks


def forward_train(model, input_ids, codec_input_ids, attention_mask, codec_labels, aux_balance_weight=None, position_ids=None,
                  return_routing: bool = False):
    """loss = sum_c CE_c(shifted codec logits) + cur_aux_weight * mean(layer aux)  (model.py:817-854), differentiable."""
    dev = model.device
    ids_dev = input_ids.to(dev)
    if codec_input_ids is not None:      # placeholder positions are replaced below: their text-table rows get no gradient (exact zeros anyway)
        ids_dev = torch.where(ids_dev == model.codec_placeholder_value, torch.full_like(ids_dev, -1), ids_dev)
    x = EmbedFn.apply(ids_dev, model.language_model.embed_tokens.weight)
    if codec_input_ids is not None:
        ci = codec_input_ids.to(dev)
        # model.py:655-661 (sum of the per-channel embedding gathers; ids are clamped into the table like the decode kernel does)
        ce = CodecEmbedFn.apply(ci.reshape(-1, model.num_channels), *[e.weight for e in model.codec_embed_tokens]).reshape(*ci.shape[:-1], -1)
        m = (input_ids.to(dev) == model.codec_placeholder_value).unsqueeze(-1).expand_as(x)
        x = x.masked_scatter(m, ce.to(x.dtype))
    pm = None
    if attention_mask is not None:
        attention_mask = attention_mask.to(dev)
        if aux_balance_weight is not None:
            aux_balance_weight = attention_mask * aux_balance_weight.to(dev)
        pm = attention_mask.bool()
    hs, auxes, routing = text_forward_train(model, x, attention_mask, position_ids, pm, aux_balance_weight)
    B, T, D = hs.shape
    Cc, V = model.num_channels, model.codec_vocab_size
    logits = ops_f32_head(hs.reshape(B * T, D).contiguous(), model.codec_head.weight).view(B, T, Cc, V)
    sl = logits[:, :-1].reshape(B * (T - 1), Cc, V).contiguous()
    lab = codec_labels.to(dev)[:, 1:].reshape(B * (T - 1), Cc).contiguous()
    codec_loss = CodecCEFn.apply(sl, lab)
    aux_mean = torch.stack([a_.float() for a_ in auxes]).mean()
    loss = codec_loss + model.cur_aux_weight * aux_mean
    model.training_steps = getattr(model, "training_steps", 0) + 1      # model.py:827 (weight read before the increment)
    return (loss, codec_loss, aux_mean, routing) if return_routing else (loss, codec_loss, aux_mean)


class _HeadFn(torch.autograd.Function):
    """codec head: fp32 logits = bf16-rounded x W^T held in fp32 (model.py:818-819: `.float()` of the bf16 Linear output)."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return ops.tlinear(x, w, out_f32=True)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dyb = dy.to(torch.bfloat16).contiguous()
        return ops.linear_grads(_pad8(dyb), x, w)


def ops_f32_head(x, w):
    return _HeadFn.apply(x, w)


# ----------------------------------------------------------------------------------------------------------------------
# Training-side glue around the step: optimizer parameter groups.
def moe_param_groups(model, weight_decay: float = 0.0, no_decay_suffixes=("bias", "layernorm.weight", "norm.weight")):
    """The parameter groups the reference trainer hands its optimizer (UniMoEV2-Preview/training/moe_trainer.py:291-332):
    `decay_parameters` / `no_decay_parameters` (transformers' rule: no decay on biases and norm weights), then DeepSpeed's
    `split_params_into_different_moe_groups_for_optimizer` (deepspeed 0.15.1, restated): every expert parameter -- marked
    `allreduce = False` with its `group_name` by AudioExperts (core.py:401-404) -- moves into a group of its own per (original
    group, expert-parallel group name), flagged `moe: True`, so the optimizer / ZeRO reduce it over the expert-data-parallel
    group only.  Returns the list of dicts ready for torch.optim.*."""
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    is_nd = lambda n: any(n.endswith(s) for s in no_decay_suffixes)      # noqa: E731
    groups = [
        {"params": [p for n, p in named if not is_nd(n)], "weight_decay": weight_decay, "name": "decay_parameters",
         "params_source": [n for n, p in named if not is_nd(n)]},
        {"params": [p for n, p in named if is_nd(n)], "weight_decay": 0.0, "name": "no_decay_parameters",
         "params_source": [n for n, p in named if is_nd(n)]},
    ]
    is_moe = lambda p: hasattr(p, "allreduce") and not p.allreduce       # noqa: E731  (deepspeed.moe.utils.is_moe_param)
    names = sorted({p.group_name for g in groups for p in g["params"] if is_moe(p)})
    out = []
    moe_groups = []
    for g in groups:
        per = {k: {**{kk: vv for kk, vv in g.items() if kk not in ("params", "params_source", "name")}, "name": k, "moe": True, "params": [],
                   "params_source": []} for k in names}
        keep_p, keep_n = [], []
        for n, p in zip(g["params_source"], g["params"]):
            if is_moe(p):
                per[p.group_name]["params"].append(p)
                per[p.group_name]["params_source"].append(n)
            else:
                keep_p.append(p)
                keep_n.append(n)
        out.append({**g, "params": keep_p, "params_source": keep_n})
        moe_groups += [v for v in per.values() if v["params"]]
    return out + moe_groups
