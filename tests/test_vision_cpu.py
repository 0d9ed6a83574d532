"""CPU suite: the PRODUCT's integer tables of the multimodal path (unimoe_audio_amd/vision.py) against the reference's own outputs
(tests/golden/vision_tower.npz, rope_index.npz; written by oracle/gen_golden.py::gen_vision from utils/UniMoE_Audio_utils.py:786-854
and utils/UniMoE_Audio_model.py:513-652).  Bit-exact integer work."""
import pytest
import torch

from conftest import load_golden
from unimoe_audio_amd import vision as V


def test_window_index_and_rotary_positions_exact():
    g = load_golden("vision_tower.npz")
    cfg = g["cfg_json"]
    widx, cu = V.window_index(g["in_grid"], cfg["window_size"], cfg["spatial_merge_size"], cfg["patch_size"])
    assert torch.equal(widx, g["out_window_index"]) and cu == g["out_cu_window"].tolist()
    pos = V.rot_pos_ids(g["in_grid"], cfg["spatial_merge_size"])
    hd = cfg["hidden_size"] // cfg["num_heads"]
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd // 2, 2, dtype=torch.float) / (hd // 2)))
    table = torch.outer(torch.arange(int(g["in_grid"][:, 1:].max()), dtype=torch.float), inv)
    assert torch.equal(table[pos].flatten(1), g["out_rot"])


def test_rope_index_exact():
    g = load_golden("rope_index.npz")
    kw = dict(spatial_merge_size=2, tokens_per_second=2, image_token_id=301, video_token_id=302, vision_start_token_id=303)
    pos, delta = V.rope_index(g["in_ids"], g["in_image_grid"], g["in_video_grid"], g["in_second_per_grid"], g["in_mask"], **kw)
    assert torch.equal(pos, g["out_pos"]) and torch.equal(delta, g["out_delta"])
    pos, delta = V.rope_index(g["in_ids"], None, None, None, g["in_mask"], **kw)
    assert torch.equal(pos, g["out_pos_text"]) and torch.equal(delta, g["out_delta_text"])


def test_scatter_checks_counts():
    x = torch.zeros(1, 5, 4)
    ids = torch.tensor([[1, 302, 302, 2, 3]])
    out = V.scatter_vision_embeddings(x, ids, 302, torch.ones(2, 4), "Video")
    assert float(out.sum()) == 8.0 and float(out[0, 1:3].sum()) == 8.0
    with pytest.raises(ValueError, match="do not match"):
        V.scatter_vision_embeddings(x, ids, 302, torch.ones(3, 4), "Video")


def test_state_dict_names_of_the_reference_geometry():
    m = V.Qwen2_5_VisionTransformerPretrainedModel(dict(depth=2, hidden_size=1280, intermediate_size=3420, num_heads=16, in_chans=3, patch_size=14,
                                                       spatial_merge_size=2, temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1],
                                                       out_hidden_size=2048))
    sd = m.state_dict()
    assert tuple(sd["patch_embed.proj.weight"].shape) == (1280, 6, 14, 14)
    assert tuple(sd["blocks.1.attn.qkv.weight"].shape) == (3840, 1280) and tuple(sd["blocks.1.mlp.down_proj.weight"].shape) == (1280, 3420)
    assert tuple(sd["merger.mlp.0.weight"].shape) == (5120, 5120) and tuple(sd["merger.mlp.2.weight"].shape) == (2048, 5120)
    assert "merger.ln_q.weight" in sd and "blocks.0.norm2.weight" in sd
