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


def test_frames_to_patches_layout_against_a_plain_loop():
    """the processor's patch layout restated with loops: patch index = ((t * gh/m + bh) * gw/m + bw) * m*m + ih * m + iw, inside a
    patch (channel, time, row, column)"""
    torch.manual_seed(0)
    F_, H, W = 4, 56, 84
    fr = torch.rand(F_, H, W, 3)
    patches, grid = V.frames_to_patches(fr, max_pixels=10 ** 9, min_pixels=1)
    assert grid.tolist() == [2, 4, 6] and patches.shape == (48, 1176)
    x = (fr.permute(0, 3, 1, 2) - torch.tensor(V.CLIP_MEAN).view(1, 3, 1, 1)) / torch.tensor(V.CLIP_STD).view(1, 3, 1, 1)
    n = 0
    for t in range(2):
        for bh in range(2):
            for bw in range(3):
                for ih in range(2):
                    for iw in range(2):
                        r0, c0 = (bh * 2 + ih) * 14, (bw * 2 + iw) * 14
                        ref = x[2 * t:2 * t + 2, :, r0:r0 + 14, c0:c0 + 14].permute(1, 0, 2, 3).reshape(-1)
                        assert torch.allclose(patches[n], ref, atol=1e-6), n
                        n += 1
    assert V.smart_resize(480, 640, 28, 4 * 28 * 28, 64 * 28 * 28) == (168, 252)      # floor to the factor inside the 64-token budget (<= 50 176 px)
    p2, g2 = V.frames_to_patches((torch.rand(3, 3, 100, 100) * 255).to(torch.uint8))
    assert g2.tolist()[0] == 2 and p2.shape[0] == int(g2.prod())                       # odd frame count: last frame repeated
