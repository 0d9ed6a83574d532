"""CPU suite: training-side glue of SURVEY.md 8f-4 -- dense -> MoE FFN slicing init (reference
UniMoEV2-Preview/training/train_unimoev2_qwen2vl.py:155-240) and the MoE optimizer parameter groups
(UniMoEV2-Preview/training/moe_trainer.py:291-332 + deepspeed.moe.utils.split_params_into_different_moe_groups_for_optimizer)."""
import pytest
import torch

from unimoe_audio_amd import checkpoint as CK


def _dense(I=12, D=4):
    g = torch.Generator().manual_seed(0)
    return {"model.layers.0.mlp.gate_proj.weight": torch.randn(I, D, generator=g), "model.layers.0.mlp.up_proj.weight": torch.randn(I, D, generator=g),
            "model.layers.0.mlp.down_proj.weight": torch.randn(D, I, generator=g)}


def _targets(n_dyn, Id, n_fix, Is, D=4, experts=None):
    t = {}
    for e in (range(n_dyn) if experts is None else experts):
        p = f"model.layers.0.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.{e}."
        t[p + "gate_proj.weight"], t[p + "up_proj.weight"], t[p + "down_proj.weight"] = (Id, D), (Id, D), (D, Id)
    for i in range(n_fix):
        p = f"model.layers.0.mlp.fixed_real_moe.{i}."
        t[p + "gate_proj.weight"], t[p + "up_proj.weight"], t[p + "down_proj.weight"] = (Is, D), (Is, D), (D, Is)
    t["model.layers.0.mlp.gate.weight"] = (n_dyn + 1 + n_fix, D)           # not an expert tensor: left alone
    return t


def test_moe_copy_all_cuts_consecutive_slices_and_wraps():
    d = _dense()
    out = CK.init_moe_from_dense(d, _targets(6, 4, 2, 6), moe_copy="all", n_dynamic_experts=6)
    g, dn = d["model.layers.0.mlp.gate_proj.weight"], d["model.layers.0.mlp.down_proj.weight"]
    pre = "model.layers.0.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts."
    for e in range(6):                       # 12 rows / 4 per expert: offsets 0, 4, 8, then wrap to 0, 4, 8
        off = (4 * e) % 12
        assert torch.equal(out[pre + f"{e}.gate_proj.weight"], g[off: off + 4])
        assert torch.equal(out[pre + f"{e}.down_proj.weight"], dn[:, off: off + 4])
    for i in range(2):                       # shared experts: always the leading slice
        assert torch.equal(out[f"model.layers.0.mlp.fixed_real_moe.{i}.up_proj.weight"], d["model.layers.0.mlp.up_proj.weight"][:6])
        assert torch.equal(out[f"model.layers.0.mlp.fixed_real_moe.{i}.down_proj.weight"], dn[:, :6])
    assert "model.layers.0.mlp.gate.weight" not in out


def test_full_size_expert_is_a_copy_and_misfit_asserts():
    d = _dense()
    out = CK.init_moe_from_dense(d, _targets(2, 12, 0, 0), moe_copy="all", n_dynamic_experts=2)
    assert torch.equal(out["model.layers.0.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.1.gate_proj.weight"],
                       d["model.layers.0.mlp.gate_proj.weight"])
    with pytest.raises(AssertionError):      # 5 does not tile 12: the reference asserts the offset lands on the end
        CK.init_moe_from_dense(d, _targets(3, 5, 0, 0), moe_copy="all", n_dynamic_experts=3)


def test_expert_parallel_start_offset_rows_only():
    """rank r of ep starts the ROW slices at (r * experts_per_rank * I) % rows; the column path starts at 0 (reference quirk)."""
    d = _dense()
    out = CK.init_moe_from_dense(d, _targets(4, 4, 0, 0, experts=[0, 1]), moe_copy="all", n_dynamic_experts=4, ep_rank=1, ep_size=2)
    pre = "model.layers.0.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts."
    g, dn = d["model.layers.0.mlp.gate_proj.weight"], d["model.layers.0.mlp.down_proj.weight"]
    assert torch.equal(out[pre + "0.gate_proj.weight"], g[8:12]) and torch.equal(out[pre + "1.gate_proj.weight"], g[0:4])
    assert torch.equal(out[pre + "0.down_proj.weight"], dn[:, 0:4]) and torch.equal(out[pre + "1.down_proj.weight"], dn[:, 4:8])


def test_moe_copy_single_and_none():
    d = _dense()
    gen = torch.Generator().manual_seed(1)
    out = CK.init_moe_from_dense(d, _targets(3, 4, 1, 6), moe_copy="single", n_dynamic_experts=3, initializer_range=0.02, generator=gen)
    pre = "model.layers.0.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts."
    assert torch.equal(out[pre + "0.gate_proj.weight"], d["model.layers.0.mlp.gate_proj.weight"][:4])
    w1 = out[pre + "1.gate_proj.weight"]
    assert w1.shape == (4, 4) and 0 < float(w1.std()) < 0.1 and not torch.equal(w1, d["model.layers.0.mlp.gate_proj.weight"][4:8])
    assert CK.init_moe_from_dense(d, _targets(3, 4, 1, 6), moe_copy="none") == {}


def test_moe_optimizer_param_groups():
    from unimoe_audio_amd import train as TR
    from unimoe_audio_amd.config import UniMoEAudioConfig
    from unimoe_audio_amd.dcmoe import UniMoEAudioSparseMoeBlock
    cfg = UniMoEAudioConfig(hidden_size=16, dynamic_intermediate_size=8, shared_intermediate_size=8)

    class Tiny(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = UniMoEAudioSparseMoeBlock(cfg)
            self.input_layernorm = torch.nn.LayerNorm(16)
            self.proj = torch.nn.Linear(16, 16, bias=True)

    m = Tiny()
    groups = TR.moe_param_groups(m, weight_decay=0.1)
    by = {(g["name"], g.get("moe", False)): g for g in groups}
    ename = m.mlp.dynamic_real_moe.expert_group_name                      # "ep_size_1"
    assert set(by) == {("decay_parameters", False), ("no_decay_parameters", False), (ename, True)}
    moe = by[(ename, True)]
    assert len(moe["params"]) == 8 * 3 and moe["weight_decay"] == 0.1
    assert all("deepspeed_experts" in n for n in moe["params_source"])
    dense = by[("decay_parameters", False)]
    assert all("deepspeed_experts" not in n for n in dense["params_source"]) and "mlp.gate.weight" in dense["params_source"]
    nd = by[("no_decay_parameters", False)]
    assert set(nd["params_source"]) == {"input_layernorm.weight", "input_layernorm.bias", "proj.bias"} and nd["weight_decay"] == 0.0
    total = sum(len(g["params"]) for g in groups)
    assert total == len(list(m.parameters()))
    opt = torch.optim.AdamW([{k: v for k, v in g.items() if k != "params_source"} for g in groups], lr=1e-3)   # usable as is
    assert len(opt.param_groups) == 3


