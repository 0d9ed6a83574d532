"""GPU tests of the vision tower on the HIP path (unimoe_audio_amd/vision.py) against the REFERENCE's own
Qwen2_5_VisionTransformerPretrainedModel outputs (tests/golden/vision_tower.npz) and against the CPU oracle at another size."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected (-m gpu) but no GPU is visible")
    from unimoe_audio_amd import _lib
    _lib.lib()
    return torch.device("cuda:0")


def _tower(cfg, w, dev):
    from unimoe_audio_amd.vision import Qwen2_5_VisionTransformerPretrainedModel as VT
    m = VT(cfg)
    missing, unexpected = m.load_state_dict(w, strict=True), None
    return m.to(dev, torch.bfloat16).eval()


def test_vision_tower_vs_reference_fixture(dev):
    """three inputs (two clips, one frame whose sides do not fill whole windows), window and full attention blocks, K padding of the
    MLP (intermediate 348 is not a multiple of 8, like the real 3420): final embeddings within bf16 tolerance of the reference's."""
    g = load_golden("vision_tower.npz")
    cfg = g["cfg_json"]
    w = {k[2:]: v for k, v in g.items() if k.startswith("w.")}
    m = _tower(cfg, w, dev)
    y = m(g["in_x"].to(dev), g["in_grid"]).cpu().float()
    ref = g["out_y"].float()
    assert y.shape == ref.shape
    rel = float((y - ref).norm() / ref.norm())
    worst = float(((y - ref).norm(dim=-1) / ref.norm(dim=-1)).max())
    print("\\nVISION TOWER vs reference: rel", rel, "worst row", worst)
    assert rel < 2 ** -6 and worst < 2 ** -4


def test_vision_tower_vs_oracle_at_model_width(dev):
    """hidden 1280 / 16 heads of 80 / intermediate 3420 / merger to 2048 (utils/config.json vision_config), 4 blocks, one 8-frame clip of
    16 x 16 patches = the BASELINE configs[4] clip shape: 1024 patches -> 256 video tokens."""
    from oracle import vision as OV
    cfg = dict(depth=4, hidden_size=1280, intermediate_size=3420, num_heads=16, in_chans=3, patch_size=14, spatial_merge_size=2,
               temporal_patch_size=2, window_size=112, fullatt_block_indexes=[1, 3], out_hidden_size=2048)
    from unimoe_audio_amd.vision import Qwen2_5_VisionTransformerPretrainedModel as VT
    torch.manual_seed(5)
    m = VT(cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() > 1:
                p.normal_(0, 0.02)
            elif "norm" in n or "ln_q" in n:
                p.copy_(1 + 0.05 * torch.randn_like(p))
            else:
                p.normal_(0, 0.02)
    m = m.to(torch.bfloat16).eval()
    w = {k: v.clone() for k, v in m.state_dict().items()}
    grid = torch.tensor([[4, 16, 16]])
    x = torch.randn(1024, 1176).to(torch.bfloat16)
    ref = OV.vision_forward(cfg, w, x, grid).float()
    y = m.to(dev)(x.to(dev), grid).cpu().float()
    assert y.shape == ref.shape == (256, 2048)
    rel = float((y - ref).norm() / ref.norm())
    print("\\nVISION TOWER vs oracle at model width: rel", rel)
    assert rel < 2 ** -6
