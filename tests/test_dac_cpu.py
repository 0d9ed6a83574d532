"""CPU suite: the DAC container (names / geometry of the published 16 kHz model), weight discovery errors, wav io."""
import os

import pytest
import torch

from unimoe_audio_amd import dac as D


def test_dac_16khz_geometry_and_parameter_names():
    m = D.DacModel(**D.DAC_16KHZ)
    sd = m.state_dict()
    assert m.hop_length == 320 and m.latent_dim == 1024 and m.sample_rate == 16000
    assert abs(sum(v.numel() for v in sd.values()) / 1e6 - 74.18) < 0.05            # the published 16 kHz model's size
    for k, shape in {"encoder.block.0.weight_v": (64, 1, 7), "encoder.block.0.weight_g": (64, 1, 1),
                     "encoder.block.1.block.0.block.0.alpha": (1, 64, 1), "encoder.block.4.block.4.weight_v": (1024, 512, 16),
                     "encoder.block.6.weight_v": (1024, 1024, 3), "quantizer.quantizers.11.codebook.weight": (1024, 8),
                     "quantizer.quantizers.0.in_proj.weight_v": (8, 1024, 1), "quantizer.quantizers.0.out_proj.weight_v": (1024, 8, 1),
                     "decoder.model.0.weight_v": (1536, 1024, 7), "decoder.model.1.block.1.weight_v": (1536, 768, 16),
                     "decoder.model.1.block.1.weight_g": (1536, 1, 1), "decoder.model.2.block.1.weight_v": (768, 384, 10),
                     "decoder.model.4.block.4.block.3.weight_v": (96, 96, 1), "decoder.model.6.weight_v": (1, 96, 7)}.items():
        assert tuple(sd[k].shape) == shape, (k, tuple(sd[k].shape))


def test_checkpoint_round_trip_by_name(tmp_path):
    m = D.DacModel(encoder_dim=8, decoder_dim=96).init_random(1)
    p = str(tmp_path / "weights_16khz.pth")
    torch.save({"state_dict": m.state_dict(), "metadata": {"kwargs": dict(encoder_dim=8, encoder_rates=[2, 4, 5, 8], decoder_dim=96,
                                                                         decoder_rates=[8, 5, 4, 2], n_codebooks=12, codebook_size=1024,
                                                                         codebook_dim=8, sample_rate=16000)}}, p)
    m2 = D.DacModel.load(p, device="cpu")
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a.float(), b), k
    sd = m.state_dict()
    sd.pop("decoder.model.6.bias")
    torch.save({"state_dict": sd}, p)
    with pytest.raises(KeyError):
        D.DacModel.load(p, device="cpu")


def test_missing_weights_raise_like_the_reference(tmp_path, monkeypatch):
    monkeypatch.delenv("DAC_WEIGHTS", raising=False)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="DAC weights not found"):
        D.Dac()


def test_wav_io_round_trip(tmp_path):
    a = torch.tensor([[0.0, 0.5, -0.5, 0.999, -1.0, 1.5]])
    p = str(tmp_path / "a.wav")
    D.write_wav_pcm16(p, a, 16000)
    b, sr = D.read_wav(p)
    assert sr == 16000 and b.shape == (1, 6)
    assert torch.allclose(b, torch.tensor([[0.0, 0.5, -0.5, 32735 / 32768, -1.0, 32767 / 32768]]), atol=1e-6)


def test_product_has_no_cpu_path():
    from unimoe_audio_amd import _lib as L
    with pytest.raises(L.UmoeError):
        D.conv1d(torch.zeros(1, 1, 8), torch.zeros(1, 1, 3), None)
