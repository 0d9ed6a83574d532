"""Drop-in mirror of the reference DCMoE module API, running on the HIP C-ABI.

Same constructor config fields, sub-module / parameter names and 6-tuple return as the reference
`UniMoEAudioSparseMoeBlock` (reference utils/UniMoE_Audio_core.py:196-358), so reference state dicts load
unchanged:
    gate.weight
    fixed_real_moe.{i}.{gate,up,down}_proj.weight
    dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.{e}.{gate,up,down}_proj.weight
forward(hidden_states[B,T,D], attention_mask[B,T] | None, aux_balance_weight[B,T] | None)
    -> (final_hidden_states, full_router_logits, dynamic_top_k, expert_mask, global_weight, aux_loss)

The arithmetic runs in libumoe_hip.so (router -> ragged dispatch -> grouped SwiGLU GEMMs -> combine); there is
no CPU implementation here: CPU tensors raise.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ep as EP
from . import ops


class _SwiGLUMLP(nn.Module):
    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.hidden_size = hidden_size
        self.intermediate_size = intermediate_size
        self.gate_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.up_proj = nn.Linear(hidden_size, intermediate_size, bias=False)
        self.down_proj = nn.Linear(intermediate_size, hidden_size, bias=False)


class AudioSharedExpertMLP(_SwiGLUMLP):
    """reference core.py:16-31"""

    def __init__(self, config):
        super().__init__(config.hidden_size, config.shared_intermediate_size)


class AudioDynamicExpertMLP(_SwiGLUMLP):
    """reference core.py:34-49"""

    def __init__(self, config):
        super().__init__(config.hidden_size, config.dynamic_intermediate_size)


class AudioExperts(nn.Module):
    """reference core.py:392-416 (parameter container; the math runs in the grouped GEMM)."""

    def __init__(self, config, num_local_experts: int, expert_group_name: Optional[str] = None):
        super().__init__()
        self.deepspeed_experts = nn.ModuleList([AudioDynamicExpertMLP(config) for _ in range(num_local_experts)])
        self.num_local_experts = num_local_experts
        for expert in self.deepspeed_experts:
            for _, p in expert.named_parameters():
                p.allreduce = False                       # core.py:401-404: reduce over the expert-DP group only
                p.group_name = expert_group_name


class AudioMOELayer(nn.Module):
    """reference core.py:419-493 (dispatch/exchange/combine live in the HIP path and unimoe_audio_amd.ep)."""

    def __init__(self, experts: nn.Module, ep_group_name, ep_size, num_local_experts: int):
        super().__init__()
        self.experts = experts
        self.ep_group = None
        self.ep_size = ep_size
        self.ep_group_name = ep_group_name
        self.num_local_experts = num_local_experts

    def _set_ep_group(self, ep_group):
        self.ep_group = ep_group


class UniMoEAudioMoE(nn.Module):
    """reference core.py:496-523"""

    def __init__(self, config, num_experts: int, ep_size: int, moe_name_prefix: str = "ep_size"):
        super().__init__()
        self.enable_expert_tensor_parallelism = getattr(config, "enable_expert_tensor_parallelism", False)
        self.ep_size = ep_size
        self.num_experts = num_experts
        self.expert_group_name = f"{moe_name_prefix}_{self.ep_size}"
        self.num_local_experts = num_experts // ep_size
        self.deepspeed_moe = AudioMOELayer(AudioExperts(config, self.num_local_experts, self.expert_group_name),
                                           self.expert_group_name, ep_size, self.num_local_experts)

    def set_deepspeed_parallelism(self, use_data_before_expert_parallel_=False, ep_group=None):
        """The reference asks DeepSpeed for the expert-parallel group (core.py:510-520); here the caller passes a
        torch.distributed group (RCCL on ROCm)."""
        self.deepspeed_moe._set_ep_group(ep_group)


TILED_MIN_ROWS = 64      # from this many rows on the expert GEMMs are compute-bound: umoe_tiled_gemm


class UniMoEAudioSparseMoeBlock(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.hidden_dim = config.hidden_size
        self.mlp_dynamic_expert_num = config.mlp_dynamic_expert_num + config.mlp_dynamic_null_expert_num
        self.mlp_dynamic_real_expert_num = config.mlp_dynamic_expert_num
        self.mlp_dynamic_null_expert_num = config.mlp_dynamic_null_expert_num
        self.mlp_dynamic_top_p = config.mlp_dynamic_top_p
        self.mlp_dynamic_top_k = config.mlp_dynamic_top_k
        self.mlp_fixed_expert_num = config.mlp_fixed_expert_num
        self.num_experts = self.mlp_dynamic_expert_num + self.mlp_fixed_expert_num
        self.ignore_differentiable_router = config.ignore_differentiable_router
        self.gate = nn.Linear(self.hidden_dim, self.num_experts, bias=False)
        self.fixed_real_moe = nn.ModuleList([AudioSharedExpertMLP(config) for _ in range(self.mlp_fixed_expert_num)])
        self.dynamic_real_moe = UniMoEAudioMoE(config, self.mlp_dynamic_real_expert_num, config.ep_size)
        self.router_jitter_noise = config.router_jitter_noise
        self.input_jitter_noise = config.input_jitter_noise
        self.min_capacity = config.min_capacity
        self.capacity_factor = config.capacity_factor
        self.token_drop = config.token_drop
        self.drop_policy = config.drop_policy
        self.avg_hidden_states_last = config.avg_hidden_states_last
        self.drop_token_num_print = config.drop_token_num_print
        self.fp32_gate = config.fp32_gate
        self.dynamic_intermediate_size = config.dynamic_intermediate_size
        self.shared_intermediate_size = config.shared_intermediate_size
        if self.drop_policy not in ("probs", "position"):
            raise ValueError(f"Invalid drop_policy: {self.drop_policy}")      # core.py:325
        self._packed = None
        self._packed_key = None

    # ---- weight pre-packing (MFMA operand order, once per weight version) -------------------------------
    def _experts(self):
        return self.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts

    def prepare(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and key == self._packed_key:
            return self._packed
        ex, sh = self._experts(), self.fixed_real_moe
        for p in self.parameters():
            if p.dtype != torch.bfloat16 or not p.is_cuda:
                raise L.UmoeError("UniMoEAudioSparseMoeBlock runs on the HIP path only: parameters must be bf16 on a ROCm "
                                  "device (call .to('cuda', torch.bfloat16)); there is no CPU implementation")
        pk = dict(
            exp_gu=[ops.pack_gate_up(m.gate_proj.weight.data, m.up_proj.weight.data) for m in ex],
            exp_dn=[ops.pack_weight(m.down_proj.weight.data) for m in ex],
            sh_gu=[ops.pack_gate_up(m.gate_proj.weight.data, m.up_proj.weight.data) for m in sh],
            sh_dn=[ops.pack_weight(m.down_proj.weight.data) for m in sh],
        )
        self._packed, self._packed_key = pk, key
        return pk

    def _mixer_noise(self, S: int, device):
        """Noise of the mixer's training branch (core.py:111-126, `training and not ignore_differentiable_router`): Gumbel(0,1)
        [S, n_dyn, n_dyn] (one row per token and round; the reference draws gumbel_rsample per k-group call) and uniform [S, n_dyn].
        `self._router_noise = (gumbel, uniform)` overrides the draw (tests inject the reference's own samples)."""
        if not (self.training and not self.ignore_differentiable_router):
            return None, None
        inj = getattr(self, "_router_noise", None)
        if inj is not None:
            return inj[0].to(device=device, dtype=torch.float32), inj[1].to(device=device, dtype=torch.float32)
        n = self.mlp_dynamic_expert_num
        u = torch.rand((S, n, n), device=device).clamp_(1e-20, 1.0 - 1e-7)
        return -torch.log(-torch.log(u)), torch.rand((S, n), device=device)

    def _input_noise(self, S: int, D: int, device):
        """Input jitter (core.py:243-244, `training and input_jitter_noise > 0`): U(1 - eps, 1 + eps) per element, fp32 [S, D].
        With the fp32 gate (core.py:240-241) the reference multiplies the FLOAT copy of the rows, which only the gate reads: the noise
        goes to the router kernel (`umoe_router_args.x_noise`), the product is never rounded to bf16.  Without it the reference's
        in-place `hidden_states *= noise` works on the caller's bf16 tensor, which `original_hidden_states` aliases: noise drawn in
        bf16, rows rounded to bf16, and the EXPERTS see the jittered rows too -- mirrored by the callers of this method.
        `self._input_noise_inject` [S, D] overrides the draw (tests inject fixed samples)."""
        if not (self.training and self.input_jitter_noise > 0):
            return None
        inj = getattr(self, "_input_noise_inject", None)
        if inj is not None:
            noise = inj.to(device=device, dtype=torch.float32).reshape(S, D).contiguous()
        else:
            noise = torch.empty((S, D), dtype=torch.float32, device=device).uniform_(1.0 - self.input_jitter_noise,
                                                                                     1.0 + self.input_jitter_noise)
        if not self.fp32_gate:
            noise = noise.to(torch.bfloat16).float()
        return noise

    # ---- forward ------------------------------------------------------------------------------------------
    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                aux_balance_weight: Optional[torch.Tensor] = None):
        if not hidden_states.is_cuda:
            raise L.UmoeError("UniMoEAudioSparseMoeBlock.forward needs device tensors; the CPU restatement lives in "
                              "oracle/ and is test infrastructure only")
        B, T, D = hidden_states.shape
        S = B * T
        x = hidden_states.reshape(S, D).contiguous()
        if x.dtype != torch.bfloat16:
            raise L.UmoeError("hidden_states must be bfloat16")
        n_dyn, n_real, n_fix = self.mlp_dynamic_expert_num, self.mlp_dynamic_real_expert_num, self.mlp_fixed_expert_num
        if torch.is_grad_enabled() and (hidden_states.requires_grad or any(p.requires_grad for p in self.parameters())):
            # training: forward + backward on the HIP kernels (shipped configuration)
            for p_ in self.parameters():
                if p_.dtype != torch.bfloat16 or not p_.is_contiguous():
                    raise L.UmoeError("training needs contiguous bfloat16 parameters")
            out, aux, logits, top_k, expert_mask, gw = _DCMoETrainFn.apply(self, x, attention_mask, aux_balance_weight, *_train_params(self))
            return out.reshape(B, T, D), logits, top_k, expert_mask, gw, aux
        pk = self.prepare()
        fp32_gate = bool(self.training and self.fp32_gate)                     # core.py:240-249
        noise = self._input_noise(S, D, x.device)                              # core.py:243-244
        if noise is not None and not fp32_gate:
            x = _jitter_rows(x, noise)                                         # in place on the aliased bf16 rows: experts see them too
        gmb, ru = self._mixer_noise(S, x.device)
        r = ops.router_fwd(x, self.gate.weight.data, n_dyn=n_dyn, n_real=n_real, n_fix=n_fix,
                           top_p=float(self.mlp_dynamic_top_p), fixed_top_k=int(self.mlp_dynamic_top_k),
                           jitter_eps=float(self.router_jitter_noise), attn_mask=attention_mask,
                           logits_bf16=not fp32_gate, gumbel=gmb, rand_u=ru, x_noise=noise if fp32_gate else None)
        logits, expert_mask = r["logits"], r["expert_mask"]
        top_k = r["top_k"] if self.mlp_dynamic_top_p != 0 else r["top_k"].to(torch.int32)
        aux = aux_loss(expert_mask, n_dyn, logits, aux_balance_weight)         # core.py:293
        global_w, moe_w = r["global_weight"], r["moe_weight"]
        if self.token_drop:                                                    # core.py:302-329
            td = ops.token_drop(logits, expert_mask, r["routing_weights"], n_dyn=n_dyn, n_real=n_real, n_fix=n_fix,
                                capacity=ops.expert_capacity(S, n_dyn, self.capacity_factor, self.min_capacity), policy=self.drop_policy)
            expert_mask, global_w, moe_w = td["expert_mask"], td["global_weight"], td["moe_weight"]
        disp = ops.dispatch_build(expert_mask, n_real)
        I_d, I_s = self.dynamic_intermediate_size, self.shared_intermediate_size
        ep = int(self.dynamic_real_moe.ep_size)
        if ep > 1:
            # ragged exchange (ep.py): every destination gets exactly its experts' slot rows; the outputs come back in THIS rank's slot
            # order, so the combine reads them with the local slot_of table as at ep_size 1
            y_back, _ = EP.ep_moe_ragged(x, disp, n_real, ep, self.dynamic_real_moe.deepspeed_moe.ep_group,
                                         lambda recv, plan: self._local_experts_ragged(recv, plan, pk))
            y_sh = self._shared_experts(x, pk) if n_fix else None
            if y_back.shape[0] == 0:
                y_back = torch.zeros((1, D), dtype=x.dtype, device=x.device)
            out = ops.combine(y_back, disp["slot_of"], moe_w, y_sh, global_w, None, n_dyn, n_fix)
        else:
            Imax = max(I_d, I_s if n_fix else 0)
            slots = S * n_real
            groups_gu, groups_dn = [], []
            for e in range(n_real):
                off, cnt = disp["offsets"][e:e + 1], disp["counts"][e:e + 1]
                groups_gu.append(dict(w=pk["exp_gu"][e], rows=disp["slot_token"], row_off=off, count=cnt,
                                      n_blocks=2 * I_d // 16, k=D))
                groups_dn.append(dict(w=pk["exp_dn"][e], row_off=off, count=cnt, n_blocks=D // 16, k=I_d))
            for i in range(n_fix):
                groups_gu.append(dict(w=pk["sh_gu"][i], static_count=S, out_row_base=slots + i * S, n_blocks=2 * I_s // 16, k=D))
                groups_dn.append(dict(w=pk["sh_dn"][i], static_count=S, a_row_base=slots + i * S, out_row_base=slots + i * S,
                                      n_blocks=D // 16, k=I_s))
            hbuf = torch.empty((slots + n_fix * S, Imax), dtype=torch.bfloat16, device=x.device)
            ybuf = torch.empty((slots + n_fix * S, D), dtype=torch.bfloat16, device=x.device)
            if S >= TILED_MIN_ROWS and n_real + n_fix <= 12:
                # many rows (prefill / training shapes): compute-bound tiled MFMA kernel straight on the nn.Linear tensors
                ex, sh = self._experts(), self.fixed_real_moe
                tg_gu, tg_dn = [], []
                for e in range(n_real):
                    off, cnt = disp["offsets"][e:e + 1], disp["counts"][e:e + 1]
                    tg_gu.append(dict(w=ex[e].gate_proj.weight.data, w2=ex[e].up_proj.weight.data, rows=disp["slot_token"],
                                      row_off=off, count=cnt))
                    tg_dn.append(dict(w=ex[e].down_proj.weight.data, row_off=off, count=cnt))
                for i in range(n_fix):
                    tg_gu.append(dict(w=sh[i].gate_proj.weight.data, w2=sh[i].up_proj.weight.data, static_count=S,
                                      out_row_base=slots + i * S))
                    tg_dn.append(dict(w=sh[i].down_proj.weight.data, static_count=S, a_row_base=slots + i * S,
                                      out_row_base=slots + i * S))
                ops.tiled_gemm(tg_gu, x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU)
                ops.tiled_gemm(tg_dn, hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16)
            else:
                ops.grouped_gemm(ops.GroupTable(groups_gu, x.device), x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, n_valid=Imax)
                ops.grouped_gemm(ops.GroupTable(groups_dn, x.device), hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16, n_valid=D)
            out = ops.combine(ybuf, disp["slot_of"], moe_w, ybuf[slots:] if n_fix else None, global_w, None, n_dyn, n_fix)
        out = out.reshape(B, T, D)
        if (not self.training) and self.avg_hidden_states_last:               # core.py:355-356
            import torch.distributed as dist
            grp = self.dynamic_real_moe.deepspeed_moe.ep_group
            if dist.is_initialized():
                dist.all_reduce(out, op=dist.ReduceOp.AVG, group=grp)
        return out, logits, top_k, expert_mask, global_w.to(hidden_states.dtype), aux

    # ---- expert-parallel pieces -----------------------------------------------------------------------------------
    def _local_experts_ragged(self, recv: torch.Tensor, plan, pk) -> torch.Tensor:
        """recv [n_recv, D]: the rows every rank routed to this rank's experts (ragged exchange, ep.RaggedPlan) -> y2 [cap2, D], the outputs
        of local expert q at rows [offsets2[q], offsets2[q] + counts2[q]) in (source rank, position) order.  The expert GEMMs gather their
        rows from `recv` through plan.list2: no re-laid copy of the received rows."""
        E_loc = len(self._experts())
        D, I_d = recv.shape[1], self.dynamic_intermediate_size
        cap2 = plan.cap2
        if recv.shape[0] == 0:
            recv = torch.zeros((1, D), dtype=recv.dtype, device=recv.device)
        hbuf = torch.empty((cap2, I_d), dtype=torch.bfloat16, device=recv.device)
        ybuf = torch.zeros((cap2, D), dtype=torch.bfloat16, device=recv.device)
        off, cnt = plan.offsets2, plan.counts2
        if cap2 >= TILED_MIN_ROWS:
            # many rows (an expert-parallel PREFILL): the tiled MFMA kernel on the local experts' own nn.Linear tensors -- per row the
            # arithmetic of the ep_size 1 block's tiled path (a row's result does not depend on which other rows share its tile)
            ex = self._experts()
            tg_gu = [dict(w=ex[q].gate_proj.weight.data, w2=ex[q].up_proj.weight.data, rows=plan.list2, row_off=off[q:q + 1], count=cnt[q:q + 1])
                     for q in range(E_loc)]
            tg_dn = [dict(w=ex[q].down_proj.weight.data, row_off=off[q:q + 1], count=cnt[q:q + 1]) for q in range(E_loc)]
            ops.tiled_gemm(tg_gu, recv, hbuf, max_rows=cap2, epilogue=ops.EPI_SWIGLU)
            ops.tiled_gemm(tg_dn, hbuf, ybuf, max_rows=cap2, epilogue=ops.EPI_BF16)
        else:
            gu = [dict(w=pk["exp_gu"][q], rows=plan.list2, row_off=off[q:q + 1], count=cnt[q:q + 1], n_blocks=2 * I_d // 16, k=D) for q in range(E_loc)]
            dn = [dict(w=pk["exp_dn"][q], row_off=off[q:q + 1], count=cnt[q:q + 1], n_blocks=D // 16, k=I_d) for q in range(E_loc)]
            ops.grouped_gemm(ops.GroupTable(gu, recv.device), recv, hbuf, max_rows=cap2, epilogue=ops.EPI_SWIGLU, n_valid=I_d)
            ops.grouped_gemm(ops.GroupTable(dn, recv.device), hbuf, ybuf, max_rows=cap2, epilogue=ops.EPI_BF16, n_valid=D)
        return ybuf

    def _shared_experts(self, x: torch.Tensor, pk) -> torch.Tensor:
        S, D = x.shape
        n_fix, I_s = self.mlp_fixed_expert_num, self.shared_intermediate_size
        gu = [dict(w=pk["sh_gu"][i], static_count=S, out_row_base=i * S, n_blocks=2 * I_s // 16, k=D) for i in range(n_fix)]
        dn = [dict(w=pk["sh_dn"][i], static_count=S, a_row_base=i * S, out_row_base=i * S, n_blocks=D // 16, k=I_s)
              for i in range(n_fix)]
        hbuf = torch.empty((n_fix * S, I_s), dtype=torch.bfloat16, device=x.device)
        ybuf = torch.empty((n_fix * S, D), dtype=torch.bfloat16, device=x.device)
        if S >= TILED_MIN_ROWS:      # (as the ep_size 1 block does from this many rows on: same kernel, same bits)
            sh = self.fixed_real_moe
            ops.tiled_gemm([dict(w=sh[i].gate_proj.weight.data, w2=sh[i].up_proj.weight.data, static_count=S, out_row_base=i * S) for i in range(n_fix)],
                           x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU)
            ops.tiled_gemm([dict(w=sh[i].down_proj.weight.data, static_count=S, a_row_base=i * S, out_row_base=i * S) for i in range(n_fix)],
                           hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16)
            return ybuf
        ops.grouped_gemm(ops.GroupTable(gu, x.device), x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, n_valid=I_s)
        ops.grouped_gemm(ops.GroupTable(dn, x.device), hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16, n_valid=D)
        return ybuf


def aux_loss(expert_mask, n_dyn, full_logits, aux_balance_weight=None):
    """reference audio_load_balancing_loss_func, core.py:361-389 (one HIP launch, fixed-order reductions)."""
    tw = None
    if aux_balance_weight is not None:
        b, t = aux_balance_weight.shape
        layers = full_logits.shape[0] // (b * t)                     # core.py:381-383
        tw = aux_balance_weight.reshape(1, b * t).expand(layers, -1).reshape(-1).to(full_logits.device)
    return ops.aux_loss(full_logits, expert_mask, n_dyn, tw)



# ------------------------------------------------------------------------------------------------------------------
def _jitter_rows(t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """bf16(float(t) * noise) for bf16 rows t [S, D] and fp32 noise [S, D] (one kernel: umoe_mul_noise)."""
    if t.numel() % 8 == 0 and t.is_contiguous():
        return ops.mul_noise(t, noise)
    return (t.float() * noise).to(torch.bfloat16)


# Training path: forward + backward of the block on the HIP kernels (torch.autograd.Function).
# Shipped configuration only (ignore_differentiable_router, no token drop, ep_size 1): gradients reach the gate through
# the softmax multipliers of the mixer's eval branch, the renormalisation and the global routing weight
# (core.py:115-119,284,178-193), exactly the graph the reference's autograd walks.
class _DCMoETrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk, x, attention_mask, aux_balance_weight, *params):
        S, D = x.shape
        n_dyn, n_real, n_fix = blk.mlp_dynamic_expert_num, blk.mlp_dynamic_real_expert_num, blk.mlp_fixed_expert_num
        I_d, I_s = blk.dynamic_intermediate_size, blk.shared_intermediate_size
        Imax = max(I_d, I_s if n_fix else 0)
        gate_w = params[0]
        ep = int(blk.dynamic_real_moe.ep_size)
        E_loc = n_real // ep                     # expert parallel: params carries this rank's experts only
        ex = [params[1 + 3 * e: 4 + 3 * e] for e in range(E_loc)]
        sh = [params[1 + 3 * E_loc + 3 * i: 4 + 3 * E_loc + 3 * i] for i in range(n_fix)]
        fp32_gate = bool(blk.training and blk.fp32_gate)
        # input jitter (core.py:243-244): multiplicative uniform noise, drawn with torch's device RNG (a stochastic regulariser: no
        # bit parity with the CPU generator exists; tests inject the samples).  fp32 gate: on the gate's float copy only, multiplied
        # inside the router kernel (x_noise); bf16 gate: on the aliased bf16 rows, so the experts see the jittered rows as well
        noise = blk._input_noise(S, D, x.device)
        ctx.jitter_all = noise is not None and not fp32_gate
        if ctx.jitter_all:
            x = _jitter_rows(x, noise)
        # buffers of the expert MLPs (the slot capacity is a host bound: S * n_real rows + alignment; what ops.dispatch_build_aligned sizes for)
        dev = x.device
        cap0 = ops._r8(S * n_real + n_real * 7)
        rows_total = cap0 + n_fix * S
        hbuf = torch.empty((rows_total, Imax), dtype=torch.bfloat16, device=dev)
        gu = torch.empty((rows_total, 2 * Imax), dtype=torch.bfloat16, device=dev)
        ybuf = torch.empty((rows_total, D), dtype=torch.bfloat16, device=dev)
        # the shared experts read x only: their two GEMMs start now, on a side stream, beside the gate / router / dispatch chain (a dozen
        # latency-bound launches the routed experts have to wait for), and are joined in front of the combine
        sh_side = ops.side_stream(dev, "shared") if (ops.bwd_overlap() and n_fix and S >= 1024) else None
        if sh_side is not None:
            sh_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(sh_side):
                ops.tiled_gemm([dict(w=sh[i][0], w2=sh[i][1], static_count=S, out_row_base=cap0 + i * S) for i in range(n_fix)], x, hbuf,
                               max_rows=S, epilogue=ops.EPI_SWIGLU, aux_out=gu)
                ops.tiled_gemm([dict(w=sh[i][2], static_count=S, a_row_base=cap0 + i * S, out_row_base=cap0 + i * S) for i in range(n_fix)], hbuf, ybuf,
                               max_rows=S, epilogue=ops.EPI_BF16)
        gmb, ru = blk._mixer_noise(S, x.device)
        r = ops.router_fwd(x, gate_w, n_dyn=n_dyn, n_real=n_real, n_fix=n_fix, top_p=float(blk.mlp_dynamic_top_p),
                           fixed_top_k=int(blk.mlp_dynamic_top_k), jitter_eps=float(blk.router_jitter_noise),
                           attn_mask=attention_mask, logits_bf16=not fp32_gate, gumbel=gmb, rand_u=ru,
                           x_noise=noise if fp32_gate else None)
        logits, mask, moe_w, global_w = r["logits"], r["expert_mask"], r["moe_weight"], r["global_weight"]
        tw = None
        if aux_balance_weight is not None:
            b, t = aux_balance_weight.shape
            tw = aux_balance_weight.reshape(1, b * t).expand(S // (b * t), -1).reshape(-1).to(x.device).float().contiguous()
        # aux loss on the mask BEFORE the drop (core.py:293): two small launches nobody in this forward waits for -- beside the expert GEMMs
        # on the side stream, joined in front of the return
        aux_side = ops.side_stream(x.device, "router") if ops.bwd_overlap() else None
        if aux_side is not None:
            aux_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(aux_side):
                aux = ops.aux_loss(logits, mask, n_dyn, tw)
        else:
            aux = ops.aux_loss(logits, mask, n_dyn, tw)
        mask0 = mask
        if blk.token_drop:                                              # core.py:302-329
            td = ops.token_drop(logits, mask, r["routing_weights"], n_dyn=n_dyn, n_real=n_real, n_fix=n_fix,
                                capacity=ops.expert_capacity(S, n_dyn, blk.capacity_factor, blk.min_capacity), policy=blk.drop_policy)
            mask, moe_w, global_w = td["expert_mask"], td["moe_weight"], td["global_weight"]
        disp = ops.dispatch_build_aligned(mask, n_real, 8)
        cap = ops._r8(disp["cap"])                       # routed slot rows [0, cap); shared expert i at cap + i*S
        assert cap == cap0
        ep = int(blk.dynamic_real_moe.ep_size)
        E_loc = n_real // ep
        ctx.ep_state = None
        if ep > 1:
            # expert parallel (core.py:455-488): this rank holds experts [rank * E_loc, (rank + 1) * E_loc); params carries those only
            ex = [params[1 + 3 * q: 4 + 3 * q] for q in range(E_loc)]
            sh = [params[1 + 3 * E_loc + 3 * i: 4 + 3 * E_loc + 3 * i] for i in range(n_fix)]
            ctx.ep_state = _ep_experts_fwd(blk, x, disp, ex, ep, E_loc, n_real, S, D, I_d, cap, ybuf)
        g_gu, g_dn = [], []
        for e in range(n_real if ep == 1 else 0):
            off, cnt = disp["offsets"][e:e + 1], disp["counts"][e:e + 1]
            g_gu.append(dict(w=ex[e][0], w2=ex[e][1], rows=disp["slot_token"], row_off=off, count=cnt))
            g_dn.append(dict(w=ex[e][2], row_off=off, count=cnt))
        for i in range(n_fix if sh_side is None else 0):
            g_gu.append(dict(w=sh[i][0], w2=sh[i][1], static_count=S, out_row_base=cap + i * S))
            g_dn.append(dict(w=sh[i][2], static_count=S, a_row_base=cap + i * S, out_row_base=cap + i * S))
        if g_gu:
            ops.tiled_gemm(g_gu, x, hbuf, max_rows=S, epilogue=ops.EPI_SWIGLU, aux_out=gu)
            ops.tiled_gemm(g_dn, hbuf, ybuf, max_rows=S, epilogue=ops.EPI_BF16)
        if sh_side is not None:
            torch.cuda.current_stream().wait_stream(sh_side)
        y_sh = ybuf[cap:] if n_fix else None
        out = ops.combine(ybuf, disp["slot_of"], moe_w, y_sh, global_w, None, n_dyn, n_fix)
        if aux_side is not None:
            torch.cuda.current_stream().wait_stream(aux_side)
            aux.record_stream(torch.cuda.current_stream())
        ctx.blk, ctx.dims = blk, (S, D, n_dyn, n_real, n_fix, I_d, I_s, Imax, cap, rows_total)
        ctx.disp, ctx.tw, ctx.noise, ctx.mask0 = disp, tw, noise, mask0
        ctx.round_factor = r.get("round_factor")
        ctx.save_for_backward(x, logits, r["sel"], r["top_k"], mask, moe_w, global_w, hbuf, gu, ybuf, *params)
        top_k = r["top_k"] if blk.mlp_dynamic_top_p != 0 else r["top_k"].to(torch.int32)
        gw_out = global_w.to(x.dtype)
        ctx.mark_non_differentiable(logits, top_k, mask, gw_out)
        return out, aux, logits, top_k, mask, gw_out

    @staticmethod
    def backward(ctx, d_out, d_aux, *unused):
        blk, disp = ctx.blk, ctx.disp
        S, D, n_dyn, n_real, n_fix, I_d, I_s, Imax, cap, rows_total = ctx.dims
        x, logits, sel, top_k, mask, moe_w, global_w, hbuf, gu, ybuf, *params = ctx.saved_tensors
        ep = int(blk.dynamic_real_moe.ep_size)
        n_loc = n_real // ep                     # experts whose parameters this rank holds
        ex = [params[1 + 3 * e: 4 + 3 * e] for e in range(n_loc)]
        sh = [params[1 + 3 * n_loc + 3 * i: 4 + 3 * n_loc + 3 * i] for i in range(n_fix)]
        dev, bf = x.device, torch.bfloat16
        if d_out is None:
            d_out = torch.zeros((S, D), dtype=bf, device=dev)
        d_out = d_out.to(bf).contiguous()
        offs, cnts = disp["offsets"], disp["counts"]
        total = offs[n_real:n_real + 1]
        # 1. combine backward: gradients of the expert outputs and of the routing weights
        dy = torch.empty((rows_total, D), dtype=bf, device=dev)
        y_sh = ybuf[cap:] if n_fix else None
        d_mw, d_gs = ops.combine_bwd(d_out, ybuf, disp["slot_of"], moe_w, y_sh, global_w, n_dyn, n_fix, dy, dy[cap:] if n_fix else None)
        grads = [None] * len(params)
        # 6. router: d(moe_w), d(shared weights), d(aux) -> d(logits) -> gate weight and input gradients.  A chain of ten small kernels
        # (one wave per token, 16-column GEMMs) that only needs step 1: it runs on a side stream beside the expert MLPs' backward (steps
        # 2.-5., GEMM bound) and is joined in front of step 7 (UMOE_BWD_OVERLAP=0: in program order on the one stream)
        noise = ctx.noise
        gate_only = noise is not None and not ctx.jitter_all               # fp32 gate: only the gate's copy was jittered
        jit = lambda t: _jitter_rows(t, noise)
        E = n_dyn + n_fix

        def router_chain():
            d_lg_aux = None
            if d_aux is not None:
                d_lg_aux = ops.aux_loss_bwd(logits, ctx.mask0, n_dyn, ctx.tw, d_aux)
            d_lg = ops.router_bwd(logits, sel, top_k, mask, d_mw, d_gs, d_lg_aux, n_dyn, n_real, n_fix, float(blk.router_jitter_noise),
                                  token_drop=bool(blk.token_drop), round_factor=ctx.round_factor)
            dl16 = torch.zeros((S, 16), dtype=bf, device=dev)
            dl16[:, :E] = d_lg.to(bf)
            dWgate = torch.empty((16, D), dtype=bf, device=dev)             # dl16^T x: the gate saw the jittered input
            ops.tiled_gemm_tn([dict(m=16, n=D, k=S)], dl16, jit(x) if gate_only else x, dWgate, k_split=-1)
            dxr = ops.linear_input_grad(dl16, params[0])                    # [S][16 >= E] x [E][D], the gate weight as stored
            return dWgate, (jit(dxr) if gate_only else dxr)

        main = torch.cuda.current_stream()
        side = ops.side_stream(dev, "router") if ops.bwd_overlap() else None
        if side is not None:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                dWgate, dx_router = router_chain()
        # 2.-5. expert MLP backward (umoe_grouped_swiglu_bwd / umoe_shared_swiglu_bwd): dH = dY Wd, SwiGLU', dX_slots, dWd/dWg/dWu
        dxe = torch.empty((rows_total, D), dtype=bf, device=dev)
        if ep > 1:
            dwg, dwu, dwd = _ep_experts_bwd(ctx.ep_state, ex, dy, dxe, disp, ep, n_loc, n_real, S, D, I_d, cap)
            for q in range(n_loc):
                grads[1 + 3 * q], grads[2 + 3 * q], grads[3 + 3 * q] = dwg[q], dwu[q], dwd[q]
        elif n_real:
            dwg, dwu, dwd = ops.experts_swiglu_bwd([tuple(ex[e]) for e in range(n_real)], x=x, h=hbuf[:cap], gu=gu[:cap], dy=dy[:cap],
                                                   dx_slots=dxe[:cap], D=D, I=I_d, max_rows=S, counts=cnts, offsets=offs,
                                                   slot_token=disp["slot_token"])
            for e in range(n_real):
                grads[1 + 3 * e], grads[2 + 3 * e], grads[3 + 3 * e] = dwg[e], dwu[e], dwd[e]
        if n_fix:
            dwg, dwu, dwd = ops.experts_swiglu_bwd([tuple(sh[i]) for i in range(n_fix)], x=x, h=hbuf[cap:], gu=gu[cap:], dy=dy[cap:],
                                                   dx_slots=dxe[cap:], D=D, I=I_s, max_rows=S, row_base=0)
            for i in range(n_fix):
                b0 = 1 + 3 * n_loc + 3 * i
                grads[b0], grads[b0 + 1], grads[b0 + 2] = dwg[i], dwu[i], dwd[i]
        if side is not None:
            main.wait_stream(side)
            dWgate.record_stream(main)          # (allocated on the side stream, read on this one from here on)
            dx_router.record_stream(main)
        else:
            dWgate, dx_router = router_chain()
        grads[0] = dWgate[:E]
        # 7. input gradient: slot rows back to tokens + shared experts + router
        dx = ops.permute_bwd(dxe, disp["slot_of"], dxe[cap:] if n_fix else None, n_fix, extra=dx_router)
        if ctx.jitter_all:
            dx = jit(dx)                                                    # every consumer read x * noise
        return (None, dx, None, None, *[gr.contiguous() if gr is not None else None for gr in grads])


# ---- expert-parallel training (core.py:455-488 under autograd: two all-to-alls forward, the same two backward) ----------------------
# RAGGED exchange (ep.py RaggedPlan): a rank sends every destination exactly the slot rows of that destination's experts (contiguous in
# the local 8-aligned dispatch order), after a fixed-size header all-to-all with the counts; the outputs -- and in the backward the input
# gradients -- come back in the owner's own slot order, so everything on the owner (combine, combine backward, permute backward) is the
# ep_size 1 code.  On the experts' rank the received rows are re-laid once into per-expert 8-aligned blocks in (source rank, position)
# order -- the order the padded exchange of round 2 produced, so the weight gradients keep their summation order.  Bytes per direction
# and layer at the training shape: ~ tokens * k / N * 4 KiB per pair of ranks instead of S-row slabs (204 MB per rank).
def _ep_experts_fwd(blk, x, disp, ex, ep, E_loc, n_real, S, D, I_d, cap, ybuf):
    grp = blk.dynamic_real_moe.deepspeed_moe.ep_group
    bf, dev = torch.bfloat16, x.device
    plan = EP.ep_ragged_plan(disp["offsets"], disp["counts"], n_real, ep, grp, align=8, device=dev)
    st = disp["slot_token"].long()
    if st.numel() < plan.n_send:
        st = torch.nn.functional.pad(st, (0, plan.n_send - st.numel()))
    xs = x[st[: plan.n_send].clamp(0, S - 1)]                                     # slot order; alignment padding rows are never used
    recv = EP.ep_exchange_rows(xs, plan.in_splits, plan.out_splits, grp)         # first all-to-all, core.py:467
    cap2 = plan.cap2
    xs2 = torch.zeros((cap2, D), dtype=bf, device=dev)
    if plan.rows2.numel():
        xs2[plan.pos2] = recv[plan.rows2]
    hbuf2 = torch.empty((cap2, I_d), dtype=bf, device=dev)
    gu2 = torch.empty((cap2, 2 * I_d), dtype=bf, device=dev)
    ybuf2 = torch.zeros((cap2, D), dtype=bf, device=dev)
    off2, cnt2 = plan.offsets2, plan.counts2
    g_gu = [dict(w=ex[q][0], w2=ex[q][1], row_off=off2[q:q + 1], count=cnt2[q:q + 1]) for q in range(E_loc)]
    g_dn = [dict(w=ex[q][2], row_off=off2[q:q + 1], count=cnt2[q:q + 1]) for q in range(E_loc)]
    ops.tiled_gemm(g_gu, xs2, hbuf2, max_rows=cap2, epilogue=ops.EPI_SWIGLU, aux_out=gu2)
    ops.tiled_gemm(g_dn, hbuf2, ybuf2, max_rows=cap2, epilogue=ops.EPI_BF16)
    ret = torch.zeros((plan.n_recv, D), dtype=bf, device=dev)
    if plan.rows2.numel():
        ret[plan.rows2] = ybuf2[plan.pos2]
    back = EP.ep_exchange_rows(ret, plan.out_splits, plan.in_splits, grp)        # second all-to-all, core.py:480: the owner's slot order
    ybuf[:cap].zero_()
    ybuf[: back.shape[0]] = back
    return dict(grp=grp, plan=plan, xs2=xs2, hbuf2=hbuf2, gu2=gu2)


def _ep_experts_bwd(st, ex, dy, dxe, disp, ep, E_loc, n_real, S, D, I_d, cap):
    bf, dev, grp, plan = torch.bfloat16, dy.device, st["grp"], st["plan"]
    cap2 = plan.cap2
    # gradients of the returned rows travel to the experts' owners (the forward's first exchange, same segments) ...
    dy_recv = EP.ep_exchange_rows(dy[:cap], plan.in_splits, plan.out_splits, grp)
    dy2 = torch.zeros((cap2, D), dtype=bf, device=dev)
    if plan.rows2.numel():
        dy2[plan.pos2] = dy_recv[plan.rows2]
    dxs2 = torch.zeros((cap2, D), dtype=bf, device=dev)
    ident = torch.arange(cap2, dtype=torch.int32, device=dev)
    dwg, dwu, dwd = ops.experts_swiglu_bwd([tuple(ex[q]) for q in range(E_loc)], x=st["xs2"], h=st["hbuf2"], gu=st["gu2"], dy=dy2,
                                           dx_slots=dxs2, D=D, I=I_d, max_rows=cap2, counts=plan.counts2, offsets=plan.offsets2,
                                           slot_token=ident)
    # ... and the gradients of the rows they received travel back to the rows' owners, into their slot order
    d_ret = torch.zeros((plan.n_recv, D), dtype=bf, device=dev)
    if plan.rows2.numel():
        d_ret[plan.rows2] = dxs2[plan.pos2]
    d_back = EP.ep_exchange_rows(d_ret, plan.out_splits, plan.in_splits, grp)
    dxe[:cap].zero_()
    dxe[: d_back.shape[0]] = d_back
    return dwg, dwu, dwd


def _train_params(blk):
    ps = [blk.gate.weight]
    for m in blk._experts():
        ps += [m.gate_proj.weight, m.up_proj.weight, m.down_proj.weight]
    for m in blk.fixed_real_moe:
        ps += [m.gate_proj.weight, m.up_proj.weight, m.down_proj.weight]
    return ps
