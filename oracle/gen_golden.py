"""Generate tests/golden/*.npz from the REAL reference (build container only).

TEST INFRASTRUCTURE ONLY.  Usage:  python -m oracle.gen_golden [--out tests/golden]
Imports /root/reference/utils/*.py through oracle/_ref_shim.py (third-party stubs only),
runs the reference's own functions on seeded inputs and stores inputs + outputs.
The fixtures are data (numbers); no reference source text is stored.

Fixture families
  router_ids_*   : reference UniMoEAudioSparseMoeBlock with an identity 11x11 gate so that
                   logits == inputs exactly; outputs k / expert_mask / global_weight / selection
                   order (core.py:236-358)
  dcmoe_*        : tiny-D full blocks, every flag combination (core.py:202-358)
  compress_*     : compress_matrix / decompress_matrix (MoE_utils.py:4-103)
  delay_*        : delay-pattern helpers, DecoderOutput, _generate_output, _preprocess_codec
  sampler_*      : _sample_next_token, T=0 and pre-multinomial probabilities
  generate_*     : the reference generate()/_decoder_step() control flow driven with a scripted
                   language model (model.py:918-1231)
  attn_*         : third-party transformers Qwen2.5-VL attention / RMSNorm / mRoPE (model.py:52-56)
"""
from __future__ import annotations

import argparse
import os
import types

import numpy as np
import torch

from . import _ref_shim


def _np(t):
    if isinstance(t, torch.Tensor):
        t = t.detach()
        if t.dtype == torch.bfloat16:
            return t.view(torch.int16).numpy().copy()      # stored as raw bits, key suffix "__bf16"
        return t.numpy().copy()
    return np.asarray(t)


def save(path, **kw):
    out = {}
    for k, v in kw.items():
        if isinstance(v, torch.Tensor) and v.dtype == torch.bfloat16:
            out[k + "__bf16"] = _np(v)
        else:
            out[k] = _np(v)
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items() if hasattr(v, "shape")} if len(out) < 12 else len(out))


class Cfg(types.SimpleNamespace):
    pass


def block_cfg(**over):
    c = Cfg(hidden_size=64, mlp_dynamic_expert_num=8, mlp_dynamic_null_expert_num=1, mlp_dynamic_top_p=0.7,
            mlp_dynamic_top_k=2, mlp_fixed_expert_num=2, ignore_differentiable_router=True, ep_size=1,
            router_jitter_noise=0.01, input_jitter_noise=0.0, min_capacity=8, capacity_factor=6.0, token_drop=False,
            drop_policy="probs", avg_hidden_states_last=False, drop_token_num_print=False, fp32_gate=True,
            dynamic_intermediate_size=96, shared_intermediate_size=64, hidden_act="silu",
            enable_expert_tensor_parallelism=False)
    c.__dict__.update(over)
    return c


class MixerRecorder:
    """Wraps the reference mixer to recover per-token selection order (core.py:262-282 groups tokens by k)."""

    def __init__(self, core):
        self.core = core
        self.orig = core.audio_sparse_expert_mixer
        self.calls = []

    def __enter__(self):
        def wrapped(scores, top_k, jitter_eps, training):
            m, s = self.orig(scores, top_k, jitter_eps, training)
            self.calls.append((top_k, s.clone(), m.clone()))
            return m, s
        self.core.audio_sparse_expert_mixer = wrapped
        return self

    def __exit__(self, *a):
        self.core.audio_sparse_expert_mixer = self.orig

    def selection(self, top_k_per_token: torch.Tensor, n_dyn: int) -> torch.Tensor:
        S = top_k_per_token.shape[0]
        sel = torch.full((S, n_dyn), -1, dtype=torch.int32)
        for k, s, _ in self.calls:
            idx = torch.nonzero(top_k_per_token == k, as_tuple=True)[0]
            sel[idx, :k] = s.to(torch.int32)
        return sel


class NoiseInjector:
    """Replaces the random draws of the reference mixer's training branch (core.py:111-126) by slices of fixed per-token tensors:
    `gumbel_rsample(shape)` -> gumbel[group rows, round], `torch.rand_like(max_scores)` -> rand[group rows, round].  The group of a
    mixer call = the tokens whose Top-P count equals the call's top_k (core.py:262-266), recovered from the captured counts."""

    def __init__(self, core, gumbel, rand):
        self.core, self.gumbel, self.rand = core, gumbel, rand
        self.rounds_served = 0

    def __enter__(self):
        core = self.core
        self.o_sel, self.o_mix, self.o_gum, self.o_rand = (core.audio_dynamic_expert_selection, core.audio_sparse_expert_mixer,
                                                           core.gumbel_rsample, torch.rand_like)
        st = {}

        def sel(logits, p):
            k = self.o_sel(logits, p)
            st["k"] = k.clone()
            return k

        def mix(scores, top_k, jitter_eps, training):
            st["idx"] = torch.nonzero(st["k"] == top_k, as_tuple=True)[0]
            st["round"] = 0
            assert st["idx"].numel() == scores.shape[0]
            return self.o_mix(scores, top_k, jitter_eps, training)

        def gum(shape, device):
            out = self.gumbel[st["idx"], st["round"]]
            assert tuple(out.shape) == tuple(shape)
            return out

        def rnd(t, *a, **k):
            out = self.rand[st["idx"], st["round"]].reshape(t.shape).to(t.dtype)
            st["round"] += 1
            self.rounds_served += 1
            return out

        core.audio_dynamic_expert_selection, core.audio_sparse_expert_mixer, core.gumbel_rsample, torch.rand_like = sel, mix, gum, rnd
        return self

    def __exit__(self, *a):
        core = self.core
        core.audio_dynamic_expert_selection, core.audio_sparse_expert_mixer, core.gumbel_rsample, torch.rand_like = (
            self.o_sel, self.o_mix, self.o_gum, self.o_rand)


class InputNoiseInjector:
    """Replaces the draw of the block's input jitter (core.py:243-244: `torch.empty_like(hidden_states).uniform_(1 - eps, 1 + eps)`) by a
    fixed tensor: `Tensor.uniform_` on a tensor of the noise's shape copies the fixed samples in (cast to that tensor's dtype -- fp32
    with the fp32 gate, bf16 without); every other `uniform_` call is left alone."""

    def __init__(self, noise):
        self.noise, self.served = noise, 0

    def __enter__(self):
        self.orig = torch.Tensor.uniform_
        inj = self

        def uni(t, *a, **k):
            if tuple(t.shape) == tuple(inj.noise.shape):
                inj.served += 1
                return t.copy_(inj.noise)
            return inj.orig(t, *a, **k)

        torch.Tensor.uniform_ = uni
        return self

    def __exit__(self, *a):
        torch.Tensor.uniform_ = self.orig


def _input_noise(c, B, T, seed):
    gn = torch.Generator().manual_seed(seed)
    eps = c.input_jitter_noise
    return (1.0 - eps) + 2.0 * eps * torch.rand((B, T, c.hidden_size), generator=gn)


def gen_router_ids(ref, out):
    core = ref.core
    torch.manual_seed(100)
    n_dyn, n_fix = 9, 2
    E = n_dyn + n_fix
    cases = {}
    S = 4096
    base = torch.randn(S, E) * 0.9
    peaky = torch.randn(S, E) * 2.7
    flat = torch.randn(S, E) * 0.05
    ties = torch.randn(S, E).mul(4).round().div(4)        # many exact ties, also in bf16
    edge = torch.zeros(64, E)
    edge[1] = 1.0
    edge[2, 0] = 30.0
    edge[3, :3] = 5.0
    edge[4] = torch.tensor([0.0, -0.0, 0, 0, 0, 0, 0, 0, 0, 1, -1])
    edge[5:] = torch.randn(59, E) * torch.logspace(-3, 1.5, 59).unsqueeze(1)
    for name, lg in dict(base=base, peaky=peaky, flat=flat, ties=ties, edge=edge).items():
        for dt in (torch.bfloat16, torch.float32):
            for top_p, top_k in ((0.7, 0), (0.0, 3), (0.5, 0), (0.9, 0)):
                if (top_p not in (0.7,)) and name not in ("base", "ties"):
                    continue
                c = block_cfg(hidden_size=E, mlp_dynamic_top_p=top_p, mlp_dynamic_top_k=top_k,
                              dynamic_intermediate_size=16, shared_intermediate_size=16)
                blk = core.UniMoEAudioSparseMoeBlock(c).eval()
                with torch.no_grad():
                    for p in blk.parameters():
                        p.normal_(0, 0.02)
                    blk.gate.weight.copy_(torch.eye(E))
                blk = blk.to(dt)
                x = lg.to(dt).unsqueeze(0)
                with MixerRecorder(core) as rec, torch.no_grad():
                    o = blk(x, None, None)
                assert torch.equal(o[1].float(), x[0].float()), "identity gate must reproduce logits exactly"
                tag = f"{name}_{'bf16' if dt == torch.bfloat16 else 'f32'}_p{top_p}_k{top_k}"
                cases[tag + "/logits"] = x[0]
                cases[tag + "/top_k"] = o[2]
                cases[tag + "/expert_mask"] = o[3]
                cases[tag + "/global_weight"] = o[4]
                cases[tag + "/sel"] = rec.selection(o[2], n_dyn)
    flatd = {}
    for k, v in cases.items():
        flatd[k.replace("/", "__")] = v
    save(os.path.join(out, "router_ids.npz"), **flatd)


def gen_dcmoe(ref, out):
    core = ref.core
    variants = dict(
        bf16_topp=dict(),
        bf16_pad=dict(_pad=True),
        f32_topp=dict(_dtype=torch.float32),
        bf16_topk2=dict(mlp_dynamic_top_p=0.0, mlp_dynamic_top_k=2),
        bf16_drop_probs=dict(token_drop=True, drop_policy="probs", capacity_factor=1.0, min_capacity=2),
        bf16_drop_pos=dict(token_drop=True, drop_policy="position", capacity_factor=1.0, min_capacity=2),
        bf16_auxw=dict(_auxw=True),
        bf16_train_fp32gate=dict(_train=True),
        bf16_d128=dict(hidden_size=128, dynamic_intermediate_size=160, shared_intermediate_size=96),
        bf16_nonull=dict(mlp_dynamic_null_expert_num=0),
        bf16_noshared=dict(mlp_fixed_expert_num=0),
        # input jitter (core.py:243-244) with its noise injected (in_input_noise): on the fp32 copy the gate reads (fp32 gate), or in
        # place on the bf16 rows that `original_hidden_states` aliases, so that the experts see them too (bf16 gate)
        bf16_train_inputjitter=dict(_train=True, _xnoise=True, input_jitter_noise=0.01),
        bf16_train_inputjitter_bf16gate=dict(_train=True, _xnoise=True, input_jitter_noise=0.01, fp32_gate=False),
    )
    for vi, (name, over) in enumerate(variants.items()):
        over = dict(over)
        pad = over.pop("_pad", False)
        dt = over.pop("_dtype", torch.bfloat16)
        auxw = over.pop("_auxw", False)
        train = over.pop("_train", False)
        xnoise = over.pop("_xnoise", False)
        c = block_cfg(**over)
        torch.manual_seed(200 + vi)
        blk = core.UniMoEAudioSparseMoeBlock(c)
        with torch.no_grad():
            for n, p in blk.named_parameters():
                p.normal_(0, 0.35 if n == "gate.weight" else 0.08)
        blk = blk.to(dt)
        blk.train(train)
        B, T = 3, 11
        x = torch.randn(B, T, c.hidden_size).to(dt)
        am = None
        if pad:
            am = torch.ones(B, T, dtype=torch.bool)
            am[0, :4] = False
            am[2, :1] = False
        aw = None
        if auxw:
            aw = torch.rand(B, T)
        xinj = InputNoiseInjector(_input_noise(c, B, T, 5000 + vi)) if xnoise else None
        if xinj is not None:
            xinj.__enter__()
        with MixerRecorder(core) as rec, torch.no_grad():
            o = blk(x.clone(), am, aw)
        if xinj is not None:
            xinj.__exit__()
            assert xinj.served == 1
        d = {"in_x": x, "out_hidden": o[0], "out_logits": o[1], "out_top_k": o[2], "out_mask": o[3],
             "out_weight": o[4], "out_aux": o[5], "out_sel": rec.selection(o[2].long(), c.mlp_dynamic_expert_num + c.mlp_dynamic_null_expert_num)}
        if am is not None:
            d["in_attention_mask"] = am
        if aw is not None:
            d["in_aux_balance_weight"] = aw
        if xinj is not None:
            d["in_input_noise"] = xinj.noise
        for n, p in blk.state_dict().items():
            d["w." + n] = p
        d["cfg_json"] = np.frombuffer(__import__("json").dumps(
            {k: v for k, v in c.__dict__.items()}).encode(), dtype=np.uint8)
        d["train"] = np.array(int(train))
        save(os.path.join(out, f"dcmoe_{name}.npz"), **d)


def gen_dcmoe_bwd(ref, out):
    """Reference block forward + autograd backward (training, shipped ignore_differentiable_router=True): gradients of
    loss = sum(out * G) + aux_coef * aux with respect to the input and every parameter."""
    core = ref.core
    variants = dict(
        train_fp32gate=dict(_train=True),
        eval_bf16gate=dict(_train=False),
        train_pad_auxw=dict(_train=True, _pad=True, _auxw=True),
        train_topk2=dict(_train=True, mlp_dynamic_top_p=0.0, mlp_dynamic_top_k=2),
        # token-drop branch in the graph (core.py:302-329): capacity = ceil(72 / 9 * 1.0) = 8 rows per column
        train_drop_probs=dict(_train=True, token_drop=True, drop_policy="probs", capacity_factor=1.0, min_capacity=2),
        train_drop_pos=dict(_train=True, token_drop=True, drop_policy="position", capacity_factor=1.0, min_capacity=2),
        # the mixer's training branch + AudioMoERoutingFunction (core.py:64-91,111-137): `ignore_differentiable_router=False`.
        # Its noise (gumbel_rsample per k-group call, torch.rand_like per round) is replaced by slices of two fixed per-token
        # tensors, saved in the fixture: in_gumbel [S, n_dyn(round), n_dyn], in_rand [S, n_dyn(round)]
        train_diffrouter=dict(_train=True, _noise=True, ignore_differentiable_router=False),
        train_diffrouter_bf16gate=dict(_train=True, _noise=True, ignore_differentiable_router=False, fp32_gate=False),
        # input jitter with injected noise (see gen_dcmoe); bf16 gate: the in-place product needs a non-leaf input (a clone), as in
        # the model, where the rows are the output of the post-attention RMSNorm
        train_inputjitter=dict(_train=True, _xnoise=True, input_jitter_noise=0.01),
        train_inputjitter_bf16gate=dict(_train=True, _xnoise=True, input_jitter_noise=0.01, fp32_gate=False),
    )
    for vi, (name, over) in enumerate(variants.items()):
        over = dict(over)
        train = over.pop("_train")
        pad = over.pop("_pad", False)
        auxw = over.pop("_auxw", False)
        noise = over.pop("_noise", False)
        xnoise = over.pop("_xnoise", False)
        c = block_cfg(**over)
        torch.manual_seed(900 + vi)
        blk = core.UniMoEAudioSparseMoeBlock(c)
        with torch.no_grad():
            for n, p in blk.named_parameters():
                p.normal_(0, 0.35 if n == "gate.weight" else 0.08)
        blk = blk.to(torch.bfloat16)
        blk.train(train)
        B, T = 3, 24
        x = torch.randn(B, T, c.hidden_size).to(torch.bfloat16).requires_grad_(True)
        G = torch.randn(B, T, c.hidden_size).to(torch.bfloat16)
        am = aw = None
        if pad:
            am = torch.ones(B, T, dtype=torch.bool)
            am[0, :5] = False
            am[2, :2] = False
        if auxw:
            aw = torch.rand(B, T)
        aux_coef = 0.3
        inj = None
        if noise:
            n_dyn_ = c.mlp_dynamic_expert_num + c.mlp_dynamic_null_expert_num
            gn = torch.Generator().manual_seed(4000 + vi)
            u = torch.rand((B * T, n_dyn_, n_dyn_), generator=gn).clamp_(1e-20, 1.0 - 1e-7)
            inj = NoiseInjector(core, -torch.log(-torch.log(u)), torch.rand((B * T, n_dyn_), generator=gn))
            inj.__enter__()
        xinj = InputNoiseInjector(_input_noise(c, B, T, 5100 + vi)) if xnoise else None
        if xinj is not None:
            xinj.__enter__()
        with MixerRecorder(core) as rec:
            o = blk(x.clone() if xnoise else x, am, aw)
        if xinj is not None:
            xinj.__exit__()
            assert xinj.served == 1
        if inj is not None:
            inj.__exit__()
        loss = (o[0].float() * G.float()).sum() + aux_coef * o[5].float()
        loss.backward()
        if c.token_drop and c.drop_policy == "probs":
            # the fixture must not hinge on torch.topk's unspecified order among equal values: the capacity-th and the next
            # selected logit of every column differ
            lg = o[1].detach().float()
            n_dyn = c.mlp_dynamic_expert_num + c.mlp_dynamic_null_expert_num
            cap = max(int(np.ceil(np.float32(B * T / n_dyn) * np.float32(c.capacity_factor))), c.min_capacity)
            assert int(o[3][:, :n_dyn].sum(0).max()) == cap, "no column is at capacity: nothing was dropped"
        d = {"in_x": x.detach(), "in_G": G, "aux_coef": np.array(aux_coef, dtype=np.float32), "out_hidden": o[0].detach(),
             "out_logits": o[1].detach(), "out_top_k": o[2], "out_mask": o[3], "out_weight": o[4].detach(), "out_aux": o[5].detach(),
             "out_sel": rec.selection(o[2].long(), c.mlp_dynamic_expert_num + c.mlp_dynamic_null_expert_num),
             "grad_x": x.grad}
        if am is not None:
            d["in_attention_mask"] = am
        if aw is not None:
            d["in_aux_balance_weight"] = aw
        if inj is not None:
            d["in_gumbel"], d["in_rand"] = inj.gumbel, inj.rand
            assert inj.rounds_served > 0
        if xinj is not None:
            d["in_input_noise"] = xinj.noise
        for n, p in blk.named_parameters():
            d["w." + n] = p.detach()
            d["g." + n] = p.grad if p.grad is not None else torch.zeros_like(p)
        d["cfg_json"] = np.frombuffer(__import__("json").dumps({k: v for k, v in c.__dict__.items()}).encode(), dtype=np.uint8)
        d["train"] = np.array(int(train))
        save(os.path.join(out, f"dcmoebwd_{name}.npz"), **d)


def gen_compress(ref, out):
    mu = ref.moe_utils
    torch.manual_seed(300)
    d = {}
    for ci, (S, E, D, p) in enumerate([(16, 8, 8, 0.45), (37, 8, 4, 0.2), (5, 3, 2, 0.0), (9, 4, 3, 1.0)]):
        mask = (torch.rand(S, E) < p).to(torch.int32)
        if ci == 1:
            mask[:, 2] = 0                                   # an expert with no tokens
        A = torch.randn(S, E, D)
        cap = int(mask.sum(0).max())
        Bc = mu.compress_matrix(A, mask, force_dim=cap, allow_larger_dim=True)
        Bm = mu.compress_matrix(mask, mask, force_dim=cap, allow_larger_dim=True)
        Ar = mu.decompress_matrix(Bc, mask, allow_larger_dim=True)
        d[f"c{ci}_A"], d[f"c{ci}_mask"], d[f"c{ci}_B"], d[f"c{ci}_Bmask"], d[f"c{ci}_Arec"] = A, mask, Bc, Bm, Ar
        d[f"c{ci}_cap"] = np.array(cap)
        if cap > 0:
            big = mu.compress_matrix(A, mask, force_dim=S + 3, allow_larger_dim=True)   # padded beyond S
            d[f"c{ci}_Bbig"] = big
    save(os.path.join(out, "compress.npz"), **d)


def _model_ns(ref, C=12):
    cfg = types.SimpleNamespace(codec_channels=C, codec_bos_value=1026, codec_eos_value=1024, codec_pad_value=1025,
                                codec_delay_pattern=[0, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18][:C],
                                codec_vocab_size=1027)
    return types.SimpleNamespace(config=cfg, device=torch.device("cpu"))


def gen_delay(ref, out):
    U = ref.utils
    torch.manual_seed(400)
    m = _model_ns(ref)
    d = {}
    prompts = [torch.randint(0, 1024, (7, 12)), None, torch.randint(0, 1024, (23, 12))]
    delayed, steps = U._prepare_audio_prompt(m, prompts)
    d["prep_delayed"], d["prep_steps"] = delayed, np.array(steps)
    for i, p in enumerate(prompts):
        d[f"prep_prompt{i}"] = p if p is not None else torch.zeros(0, 12, dtype=torch.long)
    delayed0, steps0 = U._prepare_audio_prompt(m, [None, None])
    d["prep0_delayed"], d["prep0_steps"] = delayed0, np.array(steps0)
    codes = torch.randint(0, 1027, (2, 40, 12))
    pre = U.build_delay_indices(2, 40, 12, m.config.codec_delay_pattern)
    d["delay_in"] = codes
    d["delay_out"] = U.apply_audio_delay(codes, 1025, 1026, pre)
    rpre = U.build_revert_indices(2, 40, 12, m.config.codec_delay_pattern)
    d["revert_out"] = U.revert_audio_delay(codes, 1025, rpre, 40)
    lengths = torch.tensor([15, 22])
    outs = U._generate_output(m, codes, lengths)
    d["genout_0"], d["genout_1"], d["genout_lengths"] = outs[0], outs[1], lengths
    # DecoderOutput.update_one both branches
    do = U.DecoderOutput(delayed0.clone(), steps0, torch.device("cpu"))
    upd = torch.randint(0, 1024, (2, 12))
    do.update_one(upd, 3, True)
    d["do_masked"] = do.generated_tokens.clone()
    do2 = U.DecoderOutput(delayed0.clone(), steps0, torch.device("cpu"))
    do2.update_one(upd, delayed0.shape[1], False)
    d["do_appended"], d["do_upd"] = do2.generated_tokens.clone(), upd
    if ref.mod is not None:
        codec = torch.randint(0, 1024, (9, 12)).tolist()
        pc = ref.mod.UniMoEAudio._preprocess_codec(None, codec, m.config.codec_delay_pattern, 12, 1026, 1024, 1025)
        d["pc_in"], d["pc_out"] = torch.tensor(codec), pc
    save(os.path.join(out, "delay.npz"), **d)


def gen_sampler(ref, out):
    M = ref.model.UniAudioRVQQwen2_5VLMoEForConditionalGeneration
    torch.manual_seed(500)
    d = {}
    logits = torch.randn(24, 1027) * 3.0
    logits[:, 1025:] = float("-inf")
    logits[::2, 1024] = float("-inf")
    logits[3, 1024] = 50.0                               # EOS is the arg-max on this row
    logits[5, 10] = logits[5, 20] = 40.0                 # exact tie: lowest index wins
    d["logits"] = logits
    d["argmax_T0"] = M._sample_next_token(logits.clone(), 0.0, 1.0, 45, 1024)
    captured = {}
    orig = torch.multinomial

    def fake(p, num_samples=1, **kw):
        captured["p"] = p.clone()
        return torch.argmax(p, dim=-1, keepdim=True)
    for name, (T, tp, tk) in dict(a=(1.2, 0.95, 45), b=(1.0, 1.0, 45), c=(0.7, 0.5, None), d=(1.0, 1.0, 5)).items():
        torch.multinomial = fake
        try:
            M._sample_next_token(logits.clone(), T, tp, tk, 1024)
        finally:
            torch.multinomial = orig
        d[f"probs_{name}"] = captured["p"]
        d[f"params_{name}"] = np.array([T, tp, -1 if tk is None else tk], dtype=np.float64)
    save(os.path.join(out, "sampler.npz"), **d)


class ScriptedLM:
    """Deterministic stand-in for the 36-layer text model, shared (by construction, not by import) with
    tests/test_oracle_golden.py: h_t = tanh(W (x_t + 0.25 * mean_{valid past} x)) + spike."""

    def __init__(self, D, seed, spikes):
        g = torch.Generator().manual_seed(seed)
        self.W = torch.randn(D, D, generator=g) / D ** 0.5
        self.u = torch.randn(D, generator=g)
        self.spikes = spikes            # {(row, position): scale}

    def step(self, x, key_valid, pos, cache):
        # cache: running [rows, D] sum and count of valid past inputs
        rows, T, D = x.shape
        outs = []
        s, n = (torch.zeros(rows, D), torch.zeros(rows, 1)) if cache is None else cache
        L0 = key_valid.shape[1] - T
        for t in range(T):
            xt = x[:, t].float()
            valid = key_valid[:, L0 + t].float().unsqueeze(1)
            ctx = s / n.clamp(min=1)
            h = torch.tanh((xt + 0.25 * ctx) @ self.W.T)
            for (r, p), sc in self.spikes.items():
                hit = (pos[r, t] == p)
                if bool(hit):
                    h[r] = h[r] + sc * self.u
            outs.append(h)
            s = s + xt * valid
            n = n + valid
        return torch.stack(outs, 1).to(x.dtype), (s, n)


def gen_generate(ref, out):
    Model = ref.model.UniAudioRVQQwen2_5VLMoEForConditionalGeneration
    U = ref.utils
    D, C, V = 32, 12, 43          # small vocabulary: eos 40, pad 41, bos 42 (all config-driven in the reference)
    EOS, PAD, BOS = 40, 41, 42
    for case, (seed, prompts, max_tokens, min_tokens, spikes, cfg_scale) in dict(
        a=(600, [None, None], 60, 5, {(1, 16): 6.0}, 3.0),
        b=(601, [None, None, None], 48, None, {(3, 9): 8.0, (5, 30): 8.0}, 1.5),
        d=(603, [None, None], 44, 20, {}, 2.0),
        c=(602, ["p7", None], 70, 10, {(3, 33): 7.0}, 0.0),
    ).items():
        torch.manual_seed(seed)
        B = len(prompts)
        mcfg = types.SimpleNamespace(codec_channels=C, codec_bos_value=BOS, codec_eos_value=EOS,
                                     codec_pad_value=PAD, codec_vocab_size=V,
                                     codec_delay_pattern=[0, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18])
        fake = types.SimpleNamespace(config=mcfg, device=torch.device("cpu"), num_channels=C, codec_vocab_size=V,
                                     codec_placeholder_value=299)
        fake.codec_embed_tokens = torch.nn.ModuleList([torch.nn.Embedding(V, D) for _ in range(C)])
        embed = torch.nn.Embedding(320, D)
        fake.codec_head = torch.nn.Linear(D, C * V, bias=False)
        lm = ScriptedLM(D, seed + 1, spikes)
        with torch.no_grad():
            fake.codec_head.weight.normal_(0, 0.3)
            fake.codec_head.weight[EOS] = lm.u * 2.0      # channel 0, EOS column follows the spike direction

        class LMWrap:
            embed_tokens = embed

            def __call__(self, input_ids=None, attention_mask=None, position_ids=None, past_key_values=None,
                         inputs_embeds=None, **kw):
                st = getattr(past_key_values, "_scripted", None)
                T = inputs_embeds.shape[1]
                kv = attention_mask.bool()
                h, st = lm.step(inputs_embeds, kv, position_ids, st)
                past_key_values._scripted = st
                # a real layer appends K/V; `if past_key_values:` in _decoder_step (model.py:941) relies on len() > 0
                z = torch.zeros(inputs_embeds.shape[0], 1, T, 1)
                past_key_values.update(z, z, 0)
                return types.SimpleNamespace(last_hidden_state=h, past_key_values=past_key_values)
        fake.language_model = LMWrap()
        for nm in ("codec_embedding", "calculate_input_embedding", "_decoder_step", "generate"):
            setattr(fake, nm, types.MethodType(getattr(Model, nm), fake))
        fake._sample_next_token = Model._sample_next_token
        aud = [torch.randint(0, EOS, (7, C)) if p == "p7" else None for p in prompts]
        prefill, steps = U._prepare_audio_prompt(fake, aud)
        dec = U.DecoderOutput(prefill.clone(), steps, torch.device("cpu"))
        Tp = 9
        input_ids = torch.randint(0, 290, (2 * B, Tp))
        attn = torch.ones(2 * B, Tp, dtype=torch.long)
        attn[0, :3] = 0
        attn[1, :1] = 0
        # a few codec placeholders in the prompt of every row (same count per row, as the reference builds it)
        input_ids[:, -4:-1] = 299
        codec_prompt = torch.randint(0, EOS, (2 * B * 3, C))
        with torch.no_grad():
            codes, lengths = fake.generate(input_ids=input_ids, attention_mask=attn, dec_output=dec,
                                           max_tokens=max_tokens, min_tokens=min_tokens, codec_input_ids=codec_prompt,
                                           cfg_scale=cfg_scale, temperature=1.0, top_p=1.0, cfg_filter_top_k=45,
                                           eos_prob_mul_factor=0.8, do_sample=False)
        d = dict(input_ids=input_ids, attention_mask=attn, codec_prompt=codec_prompt, prefill=prefill,
                 prefill_steps=np.array(steps), out_codes=codes, out_lengths=lengths,
                 out_tokens=dec.generated_tokens, lm_W=lm.W, lm_u=lm.u,
                 spikes=np.array([[r, p, s] for (r, p), s in spikes.items()], dtype=np.float64),
                 embed=embed.weight, codec_head=fake.codec_head.weight,
                 params=np.array([max_tokens, -1 if min_tokens is None else min_tokens, cfg_scale, 0.8, V, EOS, PAD, BOS]))
        for c in range(C):
            d[f"codec_embed_{c}"] = fake.codec_embed_tokens[c].weight
        save(os.path.join(out, f"generate_{case}.npz"), **d)
        print("   lengths", lengths.tolist(), "codes", tuple(codes.shape))


def gen_attn(ref, out):
    import transformers.models.qwen2_5_vl.modeling_qwen2_5_vl as q
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLTextConfig
    from transformers.cache_utils import DynamicCache
    torch.manual_seed(700)
    kw = dict(hidden_size=128, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=1,
              intermediate_size=64, vocab_size=64, rms_norm_eps=1e-6, max_position_embeddings=4096)
    try:
        cfg = Qwen2_5_VLTextConfig(rope_parameters={"rope_type": "default", "rope_theta": 1e6,
                                                    "mrope_section": [4, 6, 6]}, **kw)
    except TypeError:
        cfg = Qwen2_5_VLTextConfig(rope_theta=1e6, rope_scaling={"type": "default", "rope_type": "default",
                                                                 "mrope_section": [4, 6, 6]}, **kw)
    cfg._attn_implementation = "eager"
    for dt_name, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        attn = q.Qwen2_5_VLAttention(cfg, 0)
        norm = q.Qwen2_5_VLRMSNorm(128, eps=1e-6)
        rot = q.Qwen2_5_VLRotaryEmbedding(config=cfg)
        with torch.no_grad():
            for p in attn.parameters():
                p.normal_(0, 0.08)
            norm.weight.normal_(1.0, 0.1)
        attn, norm = attn.to(dt).eval(), norm.to(dt)
        rows, T = 3, 10
        x = torch.randn(rows, T + 2, 128).to(dt)
        valid = torch.ones(rows, T + 2, dtype=torch.bool)
        valid[0, :3] = False
        valid[2, :1] = False
        pos_full = (valid.long().cumsum(-1) - 1).masked_fill(~valid, 1)
        # distinct streams to exercise the mRoPE section logic
        pos3_full = torch.stack([pos_full, pos_full + 2, pos_full * 2], 0)
        cache = DynamicCache()
        outs = []
        with torch.no_grad():
            for (a, b) in ((0, T), (T, T + 1), (T + 1, T + 2)):
                xs = x[:, a:b]
                pos3 = pos3_full[:, :, a:b]
                cos, sin = rot(xs, pos3)
                L = b
                qpos = torch.arange(a, b).view(1, 1, b - a, 1)
                kpos = torch.arange(L).view(1, 1, 1, L)
                allowed = (kpos <= qpos) & valid[:, :L].view(rows, 1, 1, L)
                # avoid fully-masked rows: let a padded query see itself (the reference's sdpa path un-masks them)
                allowed = allowed | (kpos == qpos)
                m4 = torch.zeros(rows, 1, b - a, L, dtype=dt).masked_fill(~allowed, torch.finfo(dt).min)
                h = norm(xs)
                o = attn(hidden_states=h, attention_mask=m4, position_ids=pos3, past_key_values=cache,
                         position_embeddings=(cos, sin))[0]
                outs.append(o)
        d = dict(x=x, valid=valid, pos3=pos3_full, norm_w=norm.weight, out_prefill=outs[0], out_step1=outs[1],
                 out_step2=outs[2], norm_out=norm(x))
        for n, p in attn.state_dict().items():
            d["w." + n] = p
        cos, sin = rot(x, pos3_full)
        d["cos3"], d["sin3"] = cos, sin
        save(os.path.join(out, f"attn_{dt_name}.npz"), **d)


class _VisionRotary453(torch.nn.Module):
    """THIRD-PARTY compatibility stub: transformers 4.53.1 `Qwen2_5_VisionRotaryEmbedding` (the version the reference pins,
    configs/enviroment.yml:178) -- forward(seqlen) = outer(arange(seqlen), inv_freq).  The 5.15.0 class installed here takes
    position ids instead, which the reference's own rot_pos_emb (utils.py:786-813) does not pass."""

    def __init__(self, dim, theta=10000.0):
        super().__init__()
        self.dim, self.theta = dim, theta

    def forward(self, seqlen):
        # fp32 like a model loaded through from_pretrained(torch_dtype=bf16): the buffer is created with an explicit float dtype
        # and never loaded from the checkpoint, so it does not follow the parameters to bf16 (a later module.to(bf16) would)
        inv_freq = 1.0 / (self.theta ** (torch.arange(0, self.dim, 2, dtype=torch.float) / self.dim))
        return torch.outer(torch.arange(int(seqlen), dtype=torch.float), inv_freq)


def gen_vision(ref, out):
    """The reference's OWN vision tower (utils/UniMoE_Audio_utils.py:585-900: Conv3D patch embedding, window index, rot_pos_emb,
    32 x Qwen2_5_VLVisionBlock, patch merger, un-permute) at a reduced size, and `get_rope_index` (model.py:513-652)."""
    U = ref.utils
    U.Qwen2_5_VisionRotaryEmbedding = _VisionRotary453
    from transformers.models.qwen2_5_vl.configuration_qwen2_5_vl import Qwen2_5_VLVisionConfig
    kw = dict(depth=3, hidden_size=160, intermediate_size=348, num_heads=2, in_channels=3, patch_size=14, spatial_merge_size=2,
              temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1], out_hidden_size=64, hidden_act="silu")
    vc = Qwen2_5_VLVisionConfig(**kw)
    vc._attn_implementation = "sdpa"
    torch.manual_seed(3100)
    m = U.Qwen2_5_VisionTransformerPretrainedModel(vc)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
            elif "norm" in n or "ln_q" in n:
                p.copy_(1 + 0.05 * torch.randn_like(p))
            else:
                p.normal_(0, 0.02)
    m = m.to(torch.bfloat16).eval()
    grid = torch.tensor([[2, 16, 16], [1, 8, 12], [1, 18, 10]])          # two clips and a frame whose sides do not fill whole windows
    N = int((grid[:, 0] * grid[:, 1] * grid[:, 2]).sum())
    x = torch.randn(N, 3 * 2 * 14 * 14).to(torch.bfloat16)
    with torch.no_grad():
        y = m(x, grid_thw=grid)
        h0 = m.patch_embed(x)
        widx, cu_win = m.get_window_index(grid)
        rot = m.rot_pos_emb(grid)
    d = {"in_x": x, "in_grid": grid, "out_y": y, "out_patch": h0, "out_window_index": widx, "out_cu_window": torch.tensor(cu_win),
         "out_rot": rot, "cfg_json": np.frombuffer(__import__("json").dumps(kw).encode(), dtype=np.uint8)}
    for n, p in m.state_dict().items():
        d["w." + n] = p
    save(os.path.join(out, "vision_tower.npz"), **d)
    # get_rope_index: an unbound call on a stand-in `self` that carries the config fields it reads (model.py:521-525, 575-579)
    cfgns = types.SimpleNamespace(vision_config=types.SimpleNamespace(spatial_merge_size=2, tokens_per_second=2), image_token_id=301,
                                  video_token_id=302, vision_start_token_id=303)
    fake = types.SimpleNamespace(config=cfgns)
    fn = ref.model.UniAudioRVQQwen2_5VLMoEForConditionalGeneration.get_rope_index
    ids = torch.randint(0, 290, (3, 60))
    am = torch.ones(3, 60, dtype=torch.long)
    am[0, :9] = 0
    am[2, :21] = 0
    # row 0: one video of grid [2, 4, 6] -> 12 tokens; row 1: an image [1, 4, 4] -> 4 tokens and a video [3, 6, 4] -> 18; row 2: text only.
    # seconds per temporal grid 2.5 and 0.5: the reference casts them to the long dtype of its index tensor (:597-601): 2 and 0
    ids[0, 20] = 303; ids[0, 21:33] = 302
    ids[1, 5] = 303; ids[1, 6:10] = 301; ids[1, 30] = 303; ids[1, 31:49] = 302
    vg, sec = torch.tensor([[2, 4, 6], [3, 6, 4]]), torch.tensor([2.5, 0.5])
    pos, delta = fn(fake, ids, torch.tensor([[1, 4, 4]]), vg, sec, am)
    pos_t, delta_t = fn(fake, ids, None, None, None, am)                     # text-only branch (:634-652)
    save(os.path.join(out, "rope_index.npz"), in_ids=ids, in_mask=am, in_image_grid=torch.tensor([[1, 4, 4]]),
         in_video_grid=vg, in_second_per_grid=sec, out_pos=pos, out_delta=delta, out_pos_text=pos_t, out_delta_text=delta_t)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                  "tests", "golden"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    ref = _ref_shim.load_reference(True)
    gens = dict(router=gen_router_ids, dcmoe=gen_dcmoe, dcmoebwd=gen_dcmoe_bwd, compress=gen_compress, delay=gen_delay,
                sampler=gen_sampler, generate=gen_generate, attn=gen_attn, vision=gen_vision)
    for k, fn in gens.items():
        if a.only and k not in a.only.split(","):
            continue
        print("==", k)
        fn(ref, a.out)


if __name__ == "__main__":
    main()
