"""Differentiable CPU restatement of the reference DCMoE block (training graph), torch autograd on CPU.

TEST INFRASTRUCTURE ONLY: imported by tests/ and bench.py's cpu_baseline leg -- never by the product package.

Why a second restatement: oracle/dcmoe.py gets its integers from the C router (oracle/router_oracle.c), which has no
gradient.  This file restates the same forward with torch ops so that autograd yields the gradients the reference's
autograd yields.  Shipped training configuration: ignore_differentiable_router=True (the mixer runs its eval branch even
in training, core.py:272), token_drop=False, ep_size=1.

Follows reference utils/UniMoE_Audio_core.py:
  gate / fp32_gate                                  :240-252
  Top-P count                                       :157-167  (no gradient: integer)
  mixer, eval branch, one round per selected expert :94-154   (threshold mask under no_grad :105-109, softmax :118-119)
  scatter of the round weights, renormalisation     :262-284
  masks (padding, shared always on)                 :286-291
  aux load-balancing loss                           :361-389
  global routing weight                             :178-193
  MoE layer (weights * mask, experts, einsum)       :446-493, experts :34-49; shared experts :16-31,344-351
Formulation differs from the reference (all tokens advance round by round instead of being grouped by k; experts gather
their rows instead of the dense [S,E,D] expansion) -- same arithmetic per token, same dtypes.
Parity pin: tests/golden/dcmoebwd_*.npz = outputs AND gradients of the real reference (oracle/gen_golden.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

EXPERT_FMT = "dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.{e}.{p}_proj.weight"
SHARED_FMT = "fixed_real_moe.{i}.{p}_proj.weight"


def _top_p_count(dyn_logits: torch.Tensor, top_p: float) -> torch.Tensor:
    with torch.no_grad():
        p = torch.softmax(dyn_logits, dim=-1)
        ps, _ = torch.sort(p, dim=-1, descending=True)
        return (~(ps.cumsum(dim=-1) >= top_p)).sum(dim=-1) + 1


class _RoutingFn(torch.autograd.Function):
    """AudioMoERoutingFunction (core.py:64-91): forward multiplier * mask_for_one; backward hands `scores` the gradient
    grad * multiplier * (onehot(selected) - masked_gates) and nothing to the other inputs."""

    @staticmethod
    def forward(ctx, scores, multiplier, selected, gates, mask_for_one):
        ctx.save_for_backward(multiplier, selected, gates)
        return multiplier * mask_for_one

    @staticmethod
    def backward(ctx, g):
        multiplier, selected, gates = ctx.saved_tensors
        g = g * multiplier
        out = gates * g.mul(-1)
        out.scatter_add_(dim=-1, index=selected, src=g)
        return out, None, None, None, None


def routing_weights_train(dyn_logits: torch.Tensor, k: torch.Tensor, jitter_eps: float, gumbel: torch.Tensor, rand: torch.Tensor):
    """The mixer's TRAINING branch (core.py:111-137) with its noise as inputs (gumbel [S, n_dyn, n_dyn], rand [S, n_dyn], one slice
    per round): round j selects arg-max(masked_gates + gumbel_j), weighs it by the softmax multiplier * mask_for_one, where
    mask_for_one = 1 if (selected == arg-max of the softmaxed gates or rand_j > 0.75) else 0.3333; gradient by _RoutingFn.
    -> (weights [S, n_dyn], selection mask [S, n_dyn] int, selection order [S, n_dyn] int32 (-1 beyond k))."""
    S, n = dyn_logits.shape
    taken = torch.zeros((S, n), dtype=torch.bool)
    w = torch.zeros_like(dyn_logits)
    order = torch.full((S, n), -1, dtype=torch.int32)
    for j in range(int(k.max()) if S else 0):
        live = (k > j).unsqueeze(-1)
        masked = dyn_logits.masked_fill(taken, float("-inf"))
        with torch.no_grad():
            mx, _ = masked.max(dim=-1, keepdim=True)
            factor = dyn_logits.abs().clamp(min=mx.abs())
            far = ((mx - dyn_logits) / factor) > (2 * jitter_eps)
        gates = masked.masked_fill(far, float("-inf"))
        sel = (gates + gumbel[:, j]).max(dim=-1)[1].unsqueeze(-1)
        gates = torch.softmax(gates, dim=-1)
        mo = gates.gather(dim=-1, index=sel)
        _, mi = gates.max(dim=-1, keepdim=True)
        one = torch.logical_or(sel == mi, rand[:, j:j + 1] > 0.75)
        one = torch.add(0.3333, one, alpha=0.6667).type_as(gates)
        mult = _RoutingFn.apply(dyn_logits, mo, sel, gates, one)
        pick = torch.zeros((S, n), dtype=torch.bool).scatter(1, sel, True) & live
        w = w + torch.where(pick, torch.zeros_like(w).scatter(1, sel, mult), torch.zeros_like(w))
        order[:, j] = torch.where(live.squeeze(-1), sel.squeeze(-1).to(torch.int32), order[:, j])
        taken = taken | pick
    return w, taken.to(torch.int32), order


def routing_weights(dyn_logits: torch.Tensor, k: torch.Tensor, jitter_eps: float, forced_set: Optional[torch.Tensor] = None):
    """-> (weights [S, n_dyn] differentiable, selection mask [S, n_dyn] int): round j acts on the tokens with k > j.
    forced_set [S, n_dyn] bool (tests only): rows whose set has exactly k members pick inside that set -- teacher forcing of
    the integer decisions when two runs are compared whose logits differ in the last bits (near-ties would flip)."""
    S, n = dyn_logits.shape
    taken = torch.zeros((S, n), dtype=torch.bool)
    w = torch.zeros_like(dyn_logits)
    use = None
    if forced_set is not None:
        use = (forced_set.sum(-1) == k).unsqueeze(-1)
    for j in range(int(k.max()) if S else 0):
        live = (k > j).unsqueeze(-1)
        masked = dyn_logits.masked_fill(taken, float("-inf"))
        with torch.no_grad():
            mx, idx = masked.max(dim=-1, keepdim=True)
            if use is not None:
                _, idx_f = masked.masked_fill(~forced_set, float("-inf")).max(dim=-1, keepdim=True)
                idx = torch.where(use & live, idx_f, idx)       # the threshold reference mx stays the row maximum
            factor = dyn_logits.abs().clamp(min=mx.abs())
            far = ((mx - dyn_logits) / factor) > (2 * jitter_eps)
        p = torch.softmax(masked.masked_fill(far, float("-inf")), dim=-1)
        pick = torch.zeros((S, n), dtype=torch.bool).scatter(1, idx, True) & live
        w = w + torch.where(pick, p, torch.zeros_like(p))
        taken = taken | pick
    return w, taken.to(torch.int32)


def aux_loss(mask: torch.Tensor, n_dyn: int, logits: torch.Tensor, aux_balance_weight: Optional[torch.Tensor]):
    lowest = torch.finfo(logits.dtype).min
    prob = torch.softmax(logits.masked_fill(mask == 0, lowest)[:, :n_dyn], dim=-1)
    m = mask[:, :n_dyn].float()
    if aux_balance_weight is None:
        return (m.mean(0) * prob.mean(0)).sum() * n_dyn
    b, t = aux_balance_weight.shape
    w = aux_balance_weight.reshape(1, b * t, 1).expand(prob.shape[0] // (b * t), -1, n_dyn).reshape(-1, n_dyn)
    return ((m * w).sum(0) / w.sum(0) * ((prob * w).sum(0) / w.sum(0))).sum() * n_dyn


def forward(cfg, weights: Dict[str, torch.Tensor], hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
            aux_balance_weight: Optional[torch.Tensor] = None, training: bool = True, forced=None, noise=None, input_noise=None):
    """-> (out [B,T,D], logits, top_k, expert_mask, global_weight, aux); differentiable in hidden_states and weights.
    noise = (gumbel, rand): required when the mixer's training branch runs (training and not ignore_differentiable_router).
    input_noise [B,T,D]: the samples of the input jitter (core.py:243-244; required when training and input_jitter_noise > 0): on the
    float copy the gate reads with the fp32 gate, on the rows themselves (bf16, experts included: the reference's in-place product
    works on the tensor `original_hidden_states` aliases) without it."""
    B, T, D = hidden_states.shape
    n_real, n_fix = cfg.mlp_dynamic_expert_num, cfg.mlp_fixed_expert_num
    n_dyn = n_real + cfg.mlp_dynamic_null_expert_num
    x = hidden_states.reshape(-1, D)
    gate_w = weights["gate.weight"]
    jitter = training and float(getattr(cfg, "input_jitter_noise", 0.0)) > 0
    if jitter:
        assert input_noise is not None, "training with input_jitter_noise > 0: pass the samples (input_noise)"
    if training and cfg.fp32_gate:
        xg = x.float()
        if jitter:
            xg = xg * input_noise.reshape(-1, D).float()
        logits = F.linear(xg, gate_w.float())
    else:
        if jitter:
            x = x * input_noise.reshape(-1, D).to(x.dtype)
        logits = F.linear(x, gate_w)
    dyn = logits[:, :n_dyn]
    if cfg.mlp_dynamic_top_p != 0:
        k = _top_p_count(dyn, float(cfg.mlp_dynamic_top_p))
    else:
        k = torch.full((dyn.shape[0],), int(cfg.mlp_dynamic_top_k), dtype=torch.int64)
    forced_set = None
    if forced is not None:                                     # (top_k [S], expert_mask [S, E]) of another run: teacher forcing
        k = forced[0].long()
        forced_set = forced[1][:, :n_dyn] != 0
    if training and not cfg.ignore_differentiable_router:
        assert noise is not None and forced is None
        rw, sel, _ = routing_weights_train(dyn, k, float(cfg.router_jitter_noise), noise[0], noise[1])
    else:
        rw, sel = routing_weights(dyn, k, float(cfg.router_jitter_noise), forced_set)
    rw = rw / (rw.sum(dim=-1, keepdim=True) + 1e-6)
    mask = torch.cat([sel, torch.zeros((sel.shape[0], n_fix), dtype=torch.int32)], dim=-1)
    if attention_mask is not None:
        mask = mask * attention_mask.reshape(-1, 1).to(mask.dtype)
    if n_fix:
        mask[:, n_dyn:] = 1
    aux = aux_loss(mask, n_dyn, logits, aux_balance_weight)
    if cfg.token_drop:                                         # core.py:302-329 (capacity :170-175)
        from .dcmoe import capacity_of, drop_keep_mask
        cap = capacity_of(x.shape[0], n_dyn, cfg.capacity_factor, cfg.min_capacity)
        mask = drop_keep_mask(logits.detach(), mask, n_dyn, cap, cfg.drop_policy)
        rw = rw.masked_fill(~(mask[:, :n_dyn].bool()), 0.0)
        rw = rw / (rw.sum(dim=-1, keepdim=True) + 1e-6)        # :328-329
    if n_fix:
        g = torch.softmax(logits.masked_fill(mask == 0, float("-inf")), dim=-1)
        gw = torch.cat([rw * g[:, :n_dyn].sum(-1, keepdim=True), g[:, n_dyn:]], dim=-1)
    else:
        gw = rw
    gw = gw.to(x.dtype)
    out = torch.zeros_like(x)
    w_moe = gw[:, :n_real] * mask[:, :n_real]
    acc = torch.zeros((x.shape[0], D), dtype=torch.float32)
    for e in range(n_real):
        rows = torch.nonzero(mask[:, e], as_tuple=True)[0]
        if rows.numel() == 0:
            continue
        xe = x[rows]
        h = F.silu(F.linear(xe, weights[EXPERT_FMT.format(e=e, p="gate")])) * F.linear(xe, weights[EXPERT_FMT.format(e=e, p="up")])
        y = F.linear(h, weights[EXPERT_FMT.format(e=e, p="down")])
        acc = acc.index_add(0, rows, w_moe[rows, e].unsqueeze(-1).float() * y.float())   # einsum "se,sem->sm": fp32 accumulate
    out = out + acc.to(x.dtype)
    for i in range(n_fix):
        h = F.silu(F.linear(x, weights[SHARED_FMT.format(i=i, p="gate")])) * F.linear(x, weights[SHARED_FMT.format(i=i, p="up")])
        y = F.linear(h, weights[SHARED_FMT.format(i=i, p="down")])
        out = out + y * gw[:, n_dyn + i].unsqueeze(-1)
    top_k = k if cfg.mlp_dynamic_top_p != 0 else k.to(torch.int32)
    return out.reshape(B, T, D), logits, top_k, mask, gw, aux
