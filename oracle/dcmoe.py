"""CPU restatement of the reference DCMoE block (torch-CPU, reference dtype discipline).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.

Follows reference utils/UniMoE_Audio_core.py:
  forward orchestration                          :236-358
  gate / fp32_gate                               :240-252
  routing (Top-P count, mixer, masks, weights)   -> oracle/router_oracle.c
  aux load-balancing loss                        :361-389
  token drop ("probs" / "position")              :302-329, capacity :170-175
  MoE layer: capacity, dense dispatch, experts, combine   :446-493, :406-416
             (dispatch/combine semantics = compress_matrix / decompress_matrix,
              utils/UniMoE_Audio_utils.py:436-523)
  shared experts                                 :344-351
Every Linear runs in the tensor's dtype (bf16 in eval) exactly as torch-CPU does it:
fp32 accumulate, one rounding per op output.  Parity pin: tests/golden/dcmoe_*.npz
(generated from the reference by oracle/gen_golden.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import router as R

EXPERT_FMT = "dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.{e}.{p}_proj.weight"
SHARED_FMT = "fixed_real_moe.{i}.{p}_proj.weight"


def swiglu_mlp(x: torch.Tensor, wg: torch.Tensor, wu: torch.Tensor, wd: torch.Tensor) -> torch.Tensor:
    """down(silu(gate x) * up x), reference core.py:30-31,48-49"""
    return F.linear(F.silu(F.linear(x, wg)) * F.linear(x, wu), wd)


def aux_loss(expert_mask: torch.Tensor, n_dyn: int, full_logits: torch.Tensor,
             aux_balance_weight: Optional[torch.Tensor]) -> torch.Tensor:
    """core.py:361-389"""
    lowest = torch.finfo(full_logits.dtype).min
    prob = torch.softmax(full_logits.masked_fill(expert_mask == 0, lowest)[:, :n_dyn], dim=-1)
    m = expert_mask[:, :n_dyn]
    if aux_balance_weight is None:
        frac = m.float().mean(dim=0)
        mean_prob = prob.mean(dim=0)
    else:
        b, t = aux_balance_weight.shape
        layers = prob.shape[0] // (b * t)
        w = aux_balance_weight[None, :, :, None].expand(layers, b, t, n_dyn).reshape(-1, n_dyn)
        frac = (m.float() * w).sum(0) / w.sum(0)
        mean_prob = (prob * w).sum(0) / w.sum(0)
    return (frac * mean_prob).sum() * n_dyn


def capacity_of(num_tokens: int, num_experts: int, capacity_factor: float, min_capacity: int) -> int:
    """core.py:170-175 (float32 tensor arithmetic in the reference)"""
    c = int(torch.ceil(torch.tensor(num_tokens / num_experts) * torch.tensor(capacity_factor)).to(torch.int64))
    return max(c, int(min_capacity))


def drop_keep_mask(logits: torch.Tensor, expert_mask: torch.Tensor, n_dyn: int, cap: int, policy: str) -> torch.Tensor:
    """Token-drop selection (core.py:305-323) with a DEFINED order among equal logits: "probs" keeps, per dynamic column, the
    `cap` selected tokens with the largest logits, lowest token index first among equals (the reference's torch.topk leaves
    that order unspecified; tests/test_oracle_golden.py checks that this equals the reference wherever the boundary is not
    tied); "position" keeps the first `cap` selected tokens of every column (cumsum over ALL columns, shared ones included)."""
    dt = expert_mask.dtype
    if policy == "position":
        loc = torch.cumsum(expert_mask, dim=0) - 1
        return (expert_mask * torch.lt(loc, cap)).to(dt)
    if policy != "probs":
        raise ValueError(f"Invalid drop_policy: {policy}")
    S = logits.shape[0]
    cap = min(cap, S)
    keep = torch.zeros_like(expert_mask)
    keep[:, n_dyn:] = 1
    idx = torch.arange(S)
    for e in range(n_dyn):
        sel = idx[expert_mask[:, e] != 0]
        if sel.numel() <= cap:
            keep[sel, e] = 1
            continue
        v = logits[sel, e].float()
        order = sorted(range(sel.numel()), key=lambda i: (-float(v[i]), int(sel[i])))
        keep[sel[torch.tensor(order[:cap])], e] = 1
    return torch.logical_and(expert_mask, keep).to(dt)


class DCMoEOracle:
    def __init__(self, cfg, weights: Dict[str, torch.Tensor], prefix: str = ""):
        self.cfg = cfg
        self.w = weights
        self.p = prefix
        self.n_dyn = cfg.mlp_dynamic_expert_num + cfg.mlp_dynamic_null_expert_num
        self.n_real = cfg.mlp_dynamic_expert_num
        self.n_fix = cfg.mlp_fixed_expert_num
        self.last = {}

    def _w(self, name):
        return self.w[self.p + name]

    # -- MoE layer, dense dispatch padded to capacity (core.py:446-493) -----------
    def moe_layer(self, x: torch.Tensor, mask: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
        S, D = x.shape
        E = self.n_real
        weight = weight * mask                                  # :447
        counts = mask.sum(dim=0)
        C = int(counts.max())                                   # :455
        out_dense = torch.zeros((S, E, D), dtype=x.dtype)
        if C > 0:
            for e in range(E):
                rows = torch.nonzero(mask[:, e], as_tuple=True)[0]
                xin = torch.zeros((C, D), dtype=x.dtype)        # padded to capacity, tail zeroed (utils.py:480-485)
                xin[: rows.numel()] = x[rows]
                y = swiglu_mlp(xin, self._w(EXPERT_FMT.format(e=e, p="gate")), self._w(EXPERT_FMT.format(e=e, p="up")),
                               self._w(EXPERT_FMT.format(e=e, p="down")))
                out_dense[rows, e] = y[: rows.numel()]          # decompress (utils.py:488-523)
        return torch.einsum("se,sem->sm", weight, out_dense)    # :488

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                aux_balance_weight: Optional[torch.Tensor] = None, training: bool = False,
                input_noise: Optional[torch.Tensor] = None):
        """input_noise [B,T,D]: the samples of the input jitter (:243-244), required when training and input_jitter_noise > 0."""
        cfg = self.cfg
        B, T, D = hidden_states.shape
        orig = hidden_states
        h = hidden_states
        if training and cfg.fp32_gate:
            h = h.float()                                        # :240-241
        if training and float(getattr(cfg, "input_jitter_noise", 0.0)) > 0:
            assert input_noise is not None, "training with input_jitter_noise > 0: pass the samples (input_noise)"
            h = h * input_noise.to(h.dtype)                      # :243-244
            if not cfg.fp32_gate:
                orig = h                                         # in place on the tensor `original_hidden_states` aliases (:238)
        h = h.reshape(-1, D)
        gate_w = self._w("gate.weight")
        if training and cfg.fp32_gate:
            logits = F.linear(h, gate_w.float())                 # :249
        else:
            logits = F.linear(h, gate_w)                         # :251
        r = R.route(logits, self.n_dyn, self.n_real, self.n_fix, float(cfg.mlp_dynamic_top_p),
                    int(cfg.mlp_dynamic_top_k), float(cfg.router_jitter_noise),
                    None if attention_mask is None else attention_mask.reshape(-1))
        top_k = r["top_k"] if cfg.mlp_dynamic_top_p != 0 else r["top_k"].to(torch.int32)   # :255-257 dtypes
        expert_mask = r["expert_mask"]
        routing_w = r["routing_weights"]
        loss = aux_loss(expert_mask, self.n_dyn, logits, aux_balance_weight)  # :293
        global_w = r["global_weight"]
        if cfg.token_drop:                                        # :302-329
            expert_mask, routing_w, global_w = self._token_drop(logits, expert_mask, routing_w, B * T)
        x = orig.reshape(-1, D)
        global_w = global_w.to(x.dtype)                          # :339
        out = torch.zeros((B * T, D), dtype=x.dtype)
        out = out + self.moe_layer(x, expert_mask[:, : self.n_real], global_w[:, : self.n_real])   # :341-342
        for i in range(self.n_fix):                              # :344-351
            y = swiglu_mlp(x, self._w(SHARED_FMT.format(i=i, p="gate")), self._w(SHARED_FMT.format(i=i, p="up")),
                           self._w(SHARED_FMT.format(i=i, p="down")))
            out = out + y * global_w[:, self.n_dyn + i].unsqueeze(-1)
        self.last = dict(sel=r["sel"], routing_weights=routing_w)
        return out.reshape(B, T, D), logits, top_k, expert_mask, global_w, loss

    __call__ = forward

    def _token_drop(self, logits, expert_mask, routing_w, num_tokens):
        cfg = self.cfg
        n_dyn = self.n_dyn
        dyn_logits = logits[:, :n_dyn]
        cap = capacity_of(num_tokens, n_dyn, cfg.capacity_factor, cfg.min_capacity)
        dt = expert_mask.dtype
        if cfg.drop_policy == "probs":                            # :305-314
            cap = min(cap, dyn_logits.shape[0])
            dm = expert_mask[:, :n_dyn].bool()
            filled = dyn_logits.masked_fill(~dm, torch.finfo(dyn_logits.dtype).min)
            _, idx = torch.topk(filled, k=cap, dim=0, sorted=False)
            keep = torch.zeros_like(expert_mask).scatter(0, idx, 1)
            keep[:, n_dyn:] = 1
            expert_mask = torch.logical_and(expert_mask, keep)
        elif cfg.drop_policy == "position":                       # :321-323
            loc = torch.cumsum(expert_mask, dim=0) - 1
            expert_mask = expert_mask * torch.lt(loc, cap)
        else:
            raise ValueError(f"Invalid drop_policy: {cfg.drop_policy}")
        expert_mask = expert_mask.to(dt)
        routing_w = routing_w.masked_fill(~(expert_mask[:, :n_dyn].bool()), 0.0)
        routing_w = routing_w / (routing_w.sum(dim=-1, keepdim=True) + 1e-6)       # :328-329
        gw = torch.softmax(logits.masked_fill(expert_mask == 0, float("-inf")), dim=-1)   # :188
        gdyn = routing_w * gw[:, :n_dyn].sum(-1, keepdim=True)
        global_w = torch.cat((gdyn, gw[:, n_dyn:]), dim=-1)
        return expert_mask, routing_w, global_w
