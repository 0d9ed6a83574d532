"""bench.py's roofline arithmetic against SURVEY.md 8(d)'s figures (CPU, no device): the algorithmic bytes of a decode step are what
`roofline.step_frac` divides by the step time, so the formula is pinned to the survey's per-layer numbers."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _cfg():
    from unimoe_audio_amd.config import UniMoEAudioConfig
    return UniMoEAudioConfig()


def test_step_bytes_matches_the_survey_per_layer_figures():
    import bench
    cfg = _cfg()
    # attention weights 9 439 744 params (incl. the q/k/v biases), shared experts 16 908 288, gate 22 528, two norms;
    # 16 908 288 params per routed expert; KV read 2 (K, V) x 2 kv heads x 128 x 2 B = 1 KiB per cached token per row per layer;
    # codec head 12 x 1027 x 2048 params
    per_layer = 2 * (9_439_744 + 16_908_288 + 22_528 + 2 * 2048)
    routed = 2 * 16_908_288
    head = 2 * 12 * 1027 * 2048
    assert abs(per_layer / 1e6 - 52.75) < 0.01 and abs(routed / 1e6 - 33.82) < 0.01
    for rows, kv, hit in ((16, 315.0, 8.0), (2, 560.5, 5.5), (16, 0.0, 0.0)):
        want = 36 * (per_layer + routed * hit + 1024 * kv * rows) + head
        assert abs(bench.step_bytes(cfg, rows, kv, hit) - want) < 1.0, (rows, kv, hit)
    # expert parallel: a rank streams its n_real / ep local experts whatever the routing hit
    want = 36 * (per_layer + routed * 8 / 4 + 1024 * 400.0 * 16) + head
    assert abs(bench.step_bytes(cfg, 16, 400.0, 8.0, ep=4) - want) < 1.0


def test_roofline_entry_of_the_fused_launch_uses_all_three_projections():
    import bench
    cfg = _cfg()
    args = types.SimpleNamespace(batch=8)
    info = dict(mean_experts_hit=8.0, prof={"gateup": (0.060, 36), "down": (0.0, 0)}, steps=20, kv_len_first=305, kv_len_end=325, dt=0.06)
    r = bench.roofline(cfg, info, args, 1)
    # gate + up + down of 8 routed and 2 shared experts, bf16: 304.3 MB
    assert r["bytes_per_launch"] == (8 * 3 * 2752 * 2048 + 2 * 3 * 1376 * 2048) * 2 == 304_349_184
    assert abs(r["achieved"] - 304_349_184 / 60e-6 / 1e9) < 0.5 and r["peak"] == 8000.0 and r["bound"] == "hbm"
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-3
    assert abs(r["step_frac"] - r["step_bytes"] / (0.06 / 20) / 1e9 / 8000.0) < 1e-3
    # two-launch form (a down-projection launch was counted): gate/up bytes only
    info["prof"]["down"] = (0.023, 36)
    r2 = bench.roofline(cfg, info, args, 1)
    assert r2["bytes_per_launch"] == (8 * 2 * 2752 * 2048 + 2 * 2 * 1376 * 2048) * 2
