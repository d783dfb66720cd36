"""Pins the oracle's DAC-VAE restatement (oracle/samaudio_oracle.py dac_encode / dac_decode) numerically against the two
in-container implementations of the same networks in Hugging Face transformers:

  * encoder: `PeAudioDacEncoder` (transformers/models/pe_audio/modeling_pe_audio.py:48-157) - the copy of the DAC-VAE
    encoder that the PE-AV checkpoints carry, with the defaults of the reference's DACVAEConfig (rates 2/8/10/12,
    hidden 64, codebook 128); its `bottleneck` 1x1 conv is the mean half of `quantizer.in_proj` (reference codec.py:67-68);
  * decoder: `DacDecoder` (transformers/models/dac/modeling_dac.py:236-264,407-441) behind `quantizer.out_proj`
    (reference codec.py:88-89).

The un-vendored `dacvae` package itself is not reachable offline, so this is the strongest pin available: the
restatement agrees with independently written code for the same topology, weight for weight (strict key mapping).
"""
import math

import pytest
import torch

from oracle import samaudio_oracle as O
from sam_audio_amd.config import DACVAEConfig
from sam_audio_amd.synthetic import init_codec_state_dict

transformers = pytest.importorskip("transformers")


def _cfg():
    # hidden = encoder_dim * 2^4 must equal latent_dim for the HF DacConfig (it derives hidden_size that way)
    return DACVAEConfig(encoder_dim=8, encoder_rates=[2, 4, 4, 2], latent_dim=128, decoder_dim=64,
                        decoder_rates=[2, 4, 4, 2], codebook_dim=16)


def _sd(cfg, seed=0):
    class Wrap:
        audio_codec = cfg
    g = torch.Generator().manual_seed(seed)
    return init_codec_state_dict(Wrap, g, torch.device("cpu"))


def _res_unit_map(src, dst):
    return {f"{dst}.snake1.alpha": f"{src}.block.0.alpha", f"{dst}.conv1.weight": f"{src}.block.1.weight",
            f"{dst}.conv1.bias": f"{src}.block.1.bias", f"{dst}.snake2.alpha": f"{src}.block.2.alpha",
            f"{dst}.conv2.weight": f"{src}.block.3.weight", f"{dst}.conv2.bias": f"{src}.block.3.bias"}


def test_dac_encoder_restatement_matches_hf_pe_audio_dac_encoder():
    from transformers.models.dac.configuration_dac import DacConfig
    from transformers.models.pe_audio.modeling_pe_audio import PeAudioDacEncoder
    cfg = _cfg()
    sd = _sd(cfg)
    hc = DacConfig(encoder_hidden_size=cfg.encoder_dim, downsampling_ratios=cfg.encoder_rates,
                   decoder_hidden_size=cfg.decoder_dim, codebook_dim=cfg.codebook_dim)
    assert hc.hidden_size == cfg.latent_dim
    enc = PeAudioDacEncoder(hc).eval()
    E = "audio_codec.encoder.block."
    kmap = {"conv1.weight": E + "0.weight", "conv1.bias": E + "0.bias", "snake1.alpha": E + "5.alpha",
            "conv2.weight": E + "6.weight", "conv2.bias": E + "6.bias"}
    for i in range(4):
        for j in range(3):
            kmap.update(_res_unit_map(f"{E}{i + 1}.block.{j}", f"block.{i}.res_unit{j + 1}"))
        kmap[f"block.{i}.snake1.alpha"] = f"{E}{i + 1}.block.3.alpha"
        kmap[f"block.{i}.conv1.weight"] = f"{E}{i + 1}.block.4.weight"
        kmap[f"block.{i}.conv1.bias"] = f"{E}{i + 1}.block.4.bias"
    enc.load_state_dict({k: sd[v] for k, v in kmap.items()}, strict=True)
    hop = cfg.hop_length
    wav = 0.3 * torch.randn(2, 1, 9 * hop, generator=torch.Generator().manual_seed(1))
    with torch.inference_mode():
        z = enc(wav)
        w_ip, b_ip = sd["audio_codec.quantizer.in_proj.weight"], sd["audio_codec.quantizer.in_proj.bias"]
        ref = torch.nn.functional.conv1d(z, w_ip[: cfg.codebook_dim], b_ip[: cfg.codebook_dim])  # hf:166,176 bottleneck
        mine = O.dac_encode(sd, cfg, wav)
    assert mine.shape == ref.shape == (2, cfg.codebook_dim, 9)
    assert (mine - ref).abs().max() < 1e-5


def test_dac_decoder_restatement_matches_hf_dac_decoder():
    from transformers.models.dac.configuration_dac import DacConfig
    from transformers.models.dac.modeling_dac import DacDecoder
    cfg = _cfg()
    sd = _sd(cfg)
    hc = DacConfig(encoder_hidden_size=cfg.encoder_dim, downsampling_ratios=cfg.decoder_rates[::-1],
                   decoder_hidden_size=cfg.decoder_dim, codebook_dim=cfg.codebook_dim)
    assert list(hc.upsampling_ratios) == cfg.decoder_rates and hc.hidden_size == cfg.latent_dim
    dec = DacDecoder(hc).eval()
    Dm = "audio_codec.decoder.model."
    kmap = {"conv1.weight": Dm + "0.weight", "conv1.bias": Dm + "0.bias", "snake1.alpha": Dm + "5.alpha",
            "conv2.weight": Dm + "6.weight", "conv2.bias": Dm + "6.bias"}
    for i in range(4):
        kmap[f"block.{i}.snake1.alpha"] = f"{Dm}{i + 1}.block.0.alpha"
        kmap[f"block.{i}.conv_t1.weight"] = f"{Dm}{i + 1}.block.1.weight"
        kmap[f"block.{i}.conv_t1.bias"] = f"{Dm}{i + 1}.block.1.bias"
        for j in range(3):
            kmap.update(_res_unit_map(f"{Dm}{i + 1}.block.{j + 2}", f"block.{i}.res_unit{j + 1}"))
    dec.load_state_dict({k: sd[v] for k, v in kmap.items()}, strict=True)
    z = torch.randn(3, cfg.codebook_dim, 7, generator=torch.Generator().manual_seed(2))
    with torch.inference_mode():
        emb = torch.nn.functional.conv1d(z, sd["audio_codec.quantizer.out_proj.weight"],
                                         sd["audio_codec.quantizer.out_proj.bias"])           # codec.py:88
        ref = dec(emb)
        mine = O.dac_decode(sd, cfg, z)
    assert mine.shape == ref.shape == (3, 1, 7 * math.prod(cfg.decoder_rates))
    assert (mine - ref).abs().max() < 1e-5
