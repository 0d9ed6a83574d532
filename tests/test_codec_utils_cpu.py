"""CPU suite: the PRODUCT's integer helpers (unimoe_audio_amd/codec_utils.py) against the reference's own outputs
(tests/golden/delay.npz, written by oracle/gen_golden.py from utils/UniMoE_Audio_utils.py:137-325 and
utils/UniMoE_Audio_mod.py:140-156 run in the build container).  Bit-exact: these are index arithmetic on int tensors."""
import types

import torch

from conftest import load_golden
from unimoe_audio_amd import codec_utils as CU


def _cfg12():
    return types.SimpleNamespace(codec_channels=12, codec_bos_value=1026, codec_eos_value=1024, codec_pad_value=1025,
                                 codec_delay_pattern=[0, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18], codec_vocab_size=1027)


def test_prepare_audio_prompt_exact():
    g, cfg = load_golden("delay.npz"), _cfg12()
    delayed, steps = CU.prepare_audio_prompt(cfg, [g["prep_prompt0"], None, g["prep_prompt2"]])
    assert delayed.dtype == torch.int32 and torch.equal(delayed, g["prep_delayed"]) and steps == g["prep_steps"].tolist()
    d0, s0 = CU.prepare_audio_prompt(cfg, [None, None])
    assert torch.equal(d0, g["prep0_delayed"]) and s0 == g["prep0_steps"].tolist()
    # an empty prompt tensor behaves like no prompt (reference utils.py:246-252)
    d1, s1 = CU.prepare_audio_prompt(cfg, [g["prep_prompt0"], g["prep_prompt1"], g["prep_prompt2"]])
    assert torch.equal(d1, g["prep_delayed"]) and s1 == g["prep_steps"].tolist()


def test_apply_and_revert_delay_exact():
    g, cfg = load_golden("delay.npz"), _cfg12()
    codes = g["delay_in"]
    B, T, C = codes.shape
    pre = CU.build_delay_indices(B, T, C, cfg.codec_delay_pattern)
    assert torch.equal(CU.apply_audio_delay(codes, 1025, 1026, pre), g["delay_out"])
    assert torch.equal(CU.apply_audio_delay(codes, 1025, 1026, delay_pattern=cfg.codec_delay_pattern), g["delay_out"])
    rpre = CU.build_revert_indices(B, T, C, cfg.codec_delay_pattern)
    assert torch.equal(CU.revert_audio_delay(codes, 1025, rpre, T), g["revert_out"])
    assert torch.equal(CU.revert_audio_delay(codes, 1025, delay_pattern=cfg.codec_delay_pattern), g["revert_out"])


def test_generate_output_exact():
    g, cfg = load_golden("delay.npz"), _cfg12()
    outs = CU.generate_output(cfg, g["delay_in"], g["genout_lengths"])
    assert len(outs) == 2 and torch.equal(outs[0], g["genout_0"]) and torch.equal(outs[1], g["genout_1"])


def test_decoder_output_update_one_both_branches():
    g = load_golden("delay.npz")
    steps = g["prep0_steps"].tolist()
    do = CU.DecoderOutput(g["prep0_delayed"].clone(), steps, torch.device("cpu"))
    do.update_one(g["do_upd"], 3, True)                       # masked update: only the -1 entries take the prediction
    assert torch.equal(do.generated_tokens, g["do_masked"])
    do2 = CU.DecoderOutput(g["prep0_delayed"].clone(), steps, torch.device("cpu"))
    do2.update_one(g["do_upd"], g["prep0_delayed"].shape[1], False)   # append
    assert torch.equal(do2.generated_tokens, g["do_appended"])
    assert torch.equal(do2.get_tokens_at(2), g["do_appended"][:, 2:3]) and torch.equal(do2.get_tokens_at(1, 4), g["do_appended"][:, 1:4])
    assert do2.get_labels_at(0) is None


def test_preprocess_codec_exact():
    g, cfg = load_golden("delay.npz"), _cfg12()
    assert torch.equal(CU.preprocess_codec(cfg, g["pc_in"]), g["pc_out"])
    assert torch.equal(CU.preprocess_codec(cfg, g["pc_in"].tolist()), g["pc_out"])
