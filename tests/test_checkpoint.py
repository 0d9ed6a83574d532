"""Checkpoint I/O, CPU only: HF shard layout + index, the reference's key conversion
(utils/UniMoE_Audio_model.py:464-467), expert-parallel re-sharding (UniMoEV2-Preview/inference/deepspeed_ep_param_aggregation.py:16-48).
Data in, data out: nothing here computes on a device."""
import json
import os

import pytest
import torch

from unimoe_audio_amd import checkpoint as CK
from unimoe_audio_amd.config import UniMoEAudioConfig
from unimoe_audio_amd.model import UniAudioRVQQwen2_5VLMoEForConditionalGeneration as Model


def _tiny(ep_size=1, seed=3):
    cfg = UniMoEAudioConfig.tiny()
    cfg.ep_size = ep_size
    m = Model(cfg)
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed)
        for p in m.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    return cfg, m.to(torch.bfloat16)


def test_convert_key_follows_the_reference_mapping():
    assert CK.convert_key("model.layers.3.mlp.gate.weight") == "language_model.layers.3.mlp.gate.weight"
    assert CK.convert_key("model.embed_tokens.weight") == "language_model.embed_tokens.weight"
    assert CK.convert_key("model.language_model.norm.weight") == "language_model.norm.weight"
    assert CK.convert_key("model.visual.blocks.0.attn.qkv.weight") == "visual.blocks.0.attn.qkv.weight"
    assert CK.convert_key("visual.merger.mlp.0.weight") == "visual.merger.mlp.0.weight"
    assert CK.convert_key("codec_head.weight") == "codec_head.weight"
    assert CK.convert_key("lm_head.weight") == "lm_head.weight"


def test_expert_ownership_matches_the_aggregation_script():
    # 8 routed experts: ep 2 -> 4 per rank, ep 8 -> one per rank (global e -> rank e // per, local e % per)
    assert [CK.ep_owner(e, 8, 2) for e in range(8)] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 0), (1, 1), (1, 2), (1, 3)]
    assert [CK.ep_owner(e, 8, 8) for e in range(8)] == [(e, 0) for e in range(8)]
    with pytest.raises(ValueError):
        CK.ep_owner(0, 8, 3)
    k = "language_model.layers.1.mlp.dynamic_real_moe.deepspeed_moe.experts.deepspeed_experts.5.up_proj.weight"
    assert CK.ep_local_key(k, 8, 1, 2) == k.replace("experts.5.", "experts.1.")
    assert CK.ep_local_key(k, 8, 0, 2) is None
    assert CK.ep_local_key(k, 8, 0, 1) == k
    shared = "language_model.layers.1.mlp.fixed_real_moe.0.up_proj.weight"
    assert CK.ep_local_key(shared, 8, 1, 2) == shared


def test_round_trip_through_sharded_reference_layout(tmp_path):
    cfg, src = _tiny()
    d = str(tmp_path / "ckpt")
    names = CK.save_checkpoint(src.state_dict(), d, max_shard_bytes=200_000)
    assert len(names) > 3 and os.path.exists(os.path.join(d, CK.INDEX_NAME))
    idx = json.load(open(os.path.join(d, CK.INDEX_NAME)))
    assert all(k.startswith(("model.", "codec_")) for k in idx["weight_map"]), "on-disk keys use the reference spelling"
    assert set(idx["weight_map"].values()) == set(names)
    dst = Model.from_pretrained(d, torch_dtype=torch.bfloat16, attn_implementation="sdpa", config=cfg)
    a, b = src.state_dict(), dst.state_dict()
    assert a.keys() == b.keys()
    for k in a:
        assert b[k].dtype == torch.bfloat16 and torch.equal(a[k], b[k]), k
    # without the index (plain shard glob) the result is the same
    os.remove(os.path.join(d, CK.INDEX_NAME))
    dst2 = Model.from_pretrained(d, config=cfg)
    assert all(torch.equal(a[k], dst2.state_dict()[k]) for k in a)


def test_cold_tensors_are_tolerated_and_hot_ones_are_not(tmp_path):
    cfg, src = _tiny()
    sd = dict(src.state_dict())
    sd["visual.patch_embed.proj.weight"] = torch.zeros(4, 4, dtype=torch.bfloat16)        # vision tower: not owned by this path
    sd["lm_head.weight"] = torch.zeros(4, 4, dtype=torch.bfloat16)
    d = str(tmp_path / "a")
    CK.save_checkpoint(sd, d)
    missing, unexpected = CK.load_checkpoint(Model(cfg).to(torch.bfloat16), d)
    assert missing == [] and unexpected == []
    del sd["language_model.layers.0.mlp.gate.weight"]
    d2 = str(tmp_path / "b")
    CK.save_checkpoint(sd, d2)
    with pytest.raises(KeyError, match="hot-path"):
        CK.load_checkpoint(Model(cfg).to(torch.bfloat16), d2)
    sd["language_model.layers.0.mlp.gate.weight"] = torch.zeros(3, 3, dtype=torch.bfloat16)
    d3 = str(tmp_path / "c")
    CK.save_checkpoint(sd, d3)
    with pytest.raises(ValueError, match="shape"):
        CK.load_checkpoint(Model(cfg).to(torch.bfloat16), d3)
    with pytest.raises(FileNotFoundError):
        CK.load_checkpoint(Model(cfg).to(torch.bfloat16), str(tmp_path / "nothing"))


@pytest.mark.parametrize("ep", [2, 4])
def test_expert_parallel_loading_and_resharding(tmp_path, ep):
    cfg, full = _tiny()
    E = cfg.mlp_dynamic_expert_num
    d = str(tmp_path / "full")
    CK.save_checkpoint(full.state_dict(), d, max_shard_bytes=300_000)
    out = str(tmp_path / "ep")
    files = CK.reshard_experts(d, out, E, ep)
    assert files == [f"model-expert_{i}-of-total_{ep}.safetensors" for i in range(ep)]
    fsd = full.state_dict()
    per = E // ep
    for r in range(ep):
        cfg_r = UniMoEAudioConfig.tiny()
        # (a) straight from the FULL checkpoint: the rank keeps its experts under local ids, everything else replicated
        m1 = Model.from_pretrained(d, config=cfg_r, ep_rank=r, ep_size=ep)
        # (b) from the per-rank files the re-sharding wrote
        cfg_r2 = UniMoEAudioConfig.tiny()
        m2 = Model.from_pretrained(out, config=cfg_r2, ep_rank=r, ep_size=ep)
        for m in (m1, m2):
            sd = m.state_dict()
            n_exp_keys = 0
            for k, v in sd.items():
                mm = CK._EXPERT_RE.match(k)
                if mm:
                    n_exp_keys += 1
                    g = r * per + int(mm.group(2))
                    assert int(mm.group(2)) < per
                    assert torch.equal(v, fsd[f"{mm.group(1)}{g}{mm.group(3)}"]), k
                else:
                    assert torch.equal(v, fsd[k]), k
            assert n_exp_keys == per * 3 * cfg.num_hidden_layers
