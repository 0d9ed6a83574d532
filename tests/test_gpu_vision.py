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


def test_generate_with_video_prompt_prefill_and_steps_vs_oracle(dev):
    """The multimodal decode path end to end at small width: vision tower -> video embeddings scattered over <|video_pad|> tokens ->
    3-D mRoPE positions (get_rope_index) -> engine prefill with explicit positions -> teacher-forced decode steps at position
    T + step + rope_delta.  Checker: oracle vision tower + oracle text model fed the reference-shaped position ids."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_engine import build, small_cfg
    from oracle import decode as OD
    from oracle import vision as OV
    from unimoe_audio_amd.codec_utils import prepare_audio_prompt
    from unimoe_audio_amd.model import DecodeEngine
    vc = dict(depth=2, hidden_size=160, intermediate_size=348, num_heads=2, in_chans=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2,
              window_size=112, fullatt_block_indexes=[1], out_hidden_size=256, tokens_per_second=2)
    cfg = small_cfg(vision_config=vc, image_token_id=301, video_token_id=302, vision_start_token_id=303, vision_end_token_id=304)
    m, w = build(cfg, 51, 0.06)
    B, T, steps, MAXT = 2, 48, 4, 44
    torch.manual_seed(52)
    ids = torch.randint(0, 290, (2 * B, T))
    am = torch.ones(2 * B, T, dtype=torch.long)
    am[0, :5] = 0
    am[2, :2] = 0
    grid = torch.tensor([[2, 4, 6]] * (2 * B))                       # every row carries one clip: 2 x 4 x 6 patches -> 12 video tokens
    for r in range(2 * B):
        ids[r, 10] = 303
        ids[r, 11:23] = 302
        ids[r, 23] = 304
    px = torch.randn(2 * B * 48, 1176).to(torch.bfloat16)
    sec = torch.tensor([2.0] * (2 * B))
    # ---- oracle
    vw = {k[len("visual."):]: v for k, v in w.items() if k.startswith("visual.")}
    emb = OV.vision_forward(vc, vw, px, grid)
    x = OD.input_embedding(cfg, w, ids, None)
    x = x.masked_scatter((ids == 302).unsqueeze(-1).expand_as(x), emb.to(x.dtype))
    pos, delta = OV.rope_index(ids, None, grid, sec, am, merge=2, tokens_per_second=2, image_token_id=301, video_token_id=302, vision_start_token_id=303)
    tm = OD.TextModelOracle(cfg, w)
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    step0 = min(psteps) - 1
    forced = torch.randint(0, 1024, (B, max(pre.shape[1], step0 + steps + 2), cfg.codec_channels)).to(torch.int32)
    keep = pre.to(torch.int32) != -1
    forced[:, : pre.shape[1]][keep] = pre.to(torch.int32)[keep]
    with torch.no_grad():
        _, cache, _ = tm.forward(x, am.bool(), pos, None)
        key_valid = am.bool()
        refs = []
        for s in range(steps):
            key_valid = torch.cat([key_valid, torch.ones((2 * B, 1), dtype=torch.bool)], -1)
            p1 = (T + s + delta).expand(-1, 1)                         # cache_position + rope_deltas (model.py:779-790), all three streams
            tok2 = forced[:, step0 + s: step0 + s + 1].long().repeat_interleave(2, dim=0)
            h, cache, _ = tm.forward(OD.codec_embedding(cfg, w, tok2), key_valid, p1, cache)
            refs.append(torch.nn.functional.linear(h, w["codec_head.weight"]).float()[:, -1])
    # ---- HIP path
    gm = m.to(dev)
    xg = gm.multimodal_embedding(ids.to(dev), None, pixel_values_videos=px.to(dev), video_grid_thw=grid)
    assert float((xg.cpu().float() - x.float()).norm() / x.float().norm()) < 2 ** -6
    pos_g, delta_g = gm.get_rope_index(ids, None, grid, sec, am)
    assert torch.equal(pos_g.cpu(), pos) and torch.equal(delta_g.cpu(), delta)
    eng = DecodeEngine(gm, B, Lmax=T + MAXT + 8, Tmax=MAXT + 64)
    eng.prefill(xg.reshape(-1, cfg.hidden_size).contiguous(), am.to(dev), position_ids=pos_g, rope_deltas=delta_g)
    eng.start_decode(forced, psteps, MAXT, 6, cfg_scale=3.0, temperature=1.0, top_p=1.0, top_k=45, eos_mul=0.8, do_sample=False)
    for s in range(steps):
        eng.step(use_graph=(s >= 2))
        got = eng.copy_buffer("logits", torch.float32, (2 * B, cfg.codec_channels * cfg.codec_vocab_size)).cpu()
        rel = (got - refs[s]).norm(dim=-1) / refs[s].norm(dim=-1)
        assert float(rel.median()) < 0.03 and float(rel.max()) < 0.25, (s, rel.tolist())
    eng.close()


def test_generate_drops_the_pixels_like_the_reference_unless_asked(dev):
    """The reference's generate() (utils/UniMoE_Audio_model.py:1109-1131) never feeds pixel_values(_videos) to the model: the pad tokens keep
    their text embeddings and the positions stay 1-D.  Default (vision_in_generate=False): the codes with pixels equal the codes without;
    vision_in_generate=True takes the vision tower + 3-D positions and (with these weights) generates other codes."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_gpu_engine import build, small_cfg
    from unimoe_audio_amd.codec_utils import DecoderOutput, prepare_audio_prompt
    vc = dict(depth=2, hidden_size=160, intermediate_size=348, num_heads=2, in_chans=3, patch_size=14, spatial_merge_size=2, temporal_patch_size=2,
              window_size=112, fullatt_block_indexes=[1], out_hidden_size=256, tokens_per_second=2)
    cfg = small_cfg(vision_config=vc, image_token_id=301, video_token_id=302, vision_start_token_id=303, vision_end_token_id=304)
    m, _ = build(cfg, 53, 0.06)
    gm = m.to(dev)
    B, T, max_tokens = 2, 40, 12
    torch.manual_seed(54)
    ids = torch.randint(0, 290, (2 * B, T))
    am = torch.ones(2 * B, T, dtype=torch.long)
    am[0, :4] = 0
    grid = torch.tensor([[2, 4, 6]] * (2 * B))
    for r in range(2 * B):
        ids[r, 10] = 303
        ids[r, 11:23] = 302
        ids[r, 23] = 304
    px = torch.randn(2 * B * 48, 1176).to(torch.bfloat16)
    sec = torch.tensor([2.0] * (2 * B))
    pre, psteps = prepare_audio_prompt(cfg, [None] * B)
    kw = dict(cfg_scale=2.0, do_sample=True, temperature=1.0, top_p=0.9, eos_prob_mul_factor=0.8, seed=5)
    outs = []
    for pix, vig in ((False, False), (True, False), (True, True)):
        dec = DecoderOutput(pre.clone(), psteps, dev)
        extra = dict(pixel_values_videos=px.to(dev), video_grid_thw=grid, second_per_grid_ts=sec) if pix else {}
        codes, lengths = gm.generate(ids, am, dec, max_tokens, 4, vision_in_generate=vig, **extra, **kw)
        # (the logits of the last decode step: a tiny random model may emit the same codes in every mode, its logits still tell the modes apart)
        lg = gm._engine.copy_buffer("logits", torch.float32, (2 * B, cfg.codec_channels * cfg.codec_vocab_size)).cpu()
        outs.append((codes.cpu(), lengths.cpu(), dec.generated_tokens.cpu(), lg))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    assert torch.equal(outs[0][3], outs[1][3])
    assert not torch.equal(outs[0][3], outs[2][3])
