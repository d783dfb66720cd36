"""Seeded synthetic checkpoints and inputs.

There are no SAM-Audio weights, config.json files or datasets reachable offline (SURVEY.md §0,
§8c/d), so tests, smoke() and bench.py run on random-init weights of the reference architecture
and on synthetic clips.  Key names follow the reference state_dict (SURVEY.md §8b "Ownership");
the ``audio_codec.*`` names follow descript-audio-codec's Sequential numbering, which is what the
un-vendored ``dacvae`` package derives from (documented assumption - see DESIGN.md).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from .config import SAMAudioConfig


def _uniform(gen, shape, bound, device):
    if gen is None:  # meta device: shapes only
        return torch.empty(shape, device=device, dtype=torch.float32)
    return (torch.rand(shape, generator=gen, device=device, dtype=torch.float32) * 2 - 1) * bound


def _normal(gen, shape, std, device):
    if gen is None:
        return torch.empty(shape, device=device, dtype=torch.float32)
    return torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std


def init_state_dict(cfg: SAMAudioConfig, seed: int = 0, device="cpu",
                    with_codec: bool = True) -> Dict[str, torch.Tensor]:
    """Random weights with the reference's init scales (Linear/Conv: U(+-1/sqrt(fan_in));
    scale_shift tables: randn/sqrt(D), reference transformer.py:350-352,469-471).  Gates are set
    to 0.5 (the reference initialises them to 0, align.py:26 / model.py:51, which would switch
    the video / anchor terms off and leave them untested)."""
    dev = torch.device(device)
    gen = None
    if dev.type != "meta":
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
    t = cfg.transformer
    D, F, hd = t.dim, t.ffn_hidden, t.head_dim
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=False, gain=1.0):
        b = gain / math.sqrt(in_f)
        sd[name + ".weight"] = _uniform(gen, (out_f, in_f), b, dev)
        if bias:
            sd[name + ".bias"] = _uniform(gen, (out_f,), b, dev)

    def norm_w(name, n):
        sd[name] = 1.0 + _normal(gen, (n,), 0.1, dev)

    P = "transformer."
    sd[P + "final_layer_scale_shift_table"] = _normal(gen, (2, D), 1 / math.sqrt(D), dev)
    for i in range(t.n_layers):
        L = f"{P}layers.{i}."
        sd[L + "scale_shift_table"] = _normal(gen, (6, D), 1 / math.sqrt(D), dev)
        for att in ("attention", "cross_attention"):
            for w in ("wq", "wk", "wv", "wo"):
                lin(f"{L}{att}.{w}", D, D)
            norm_w(f"{L}{att}.q_norm.weight", hd)
            norm_w(f"{L}{att}.k_norm.weight", hd)
        lin(L + "feed_forward.w1", F, D)
        lin(L + "feed_forward.w2", D, F)
        lin(L + "feed_forward.w3", F, D)
        norm_w(L + "attention_norm.weight", D)
        norm_w(L + "ffn_norm.weight", D)
    norm_w(P + "norm.weight", D)
    lin(P + "output", t.out_channels, D)
    for blk in ("block1", "block2"):
        B = f"{P}x_embedder.block.{blk}."
        norm_w(B + "groupnorm.weight", D)
        sd[B + "groupnorm.bias"] = _normal(gen, (D,), 0.1, dev)
        b = 1 / math.sqrt(3 * D)
        sd[B + "project.weight"] = _uniform(gen, (D, D, 3), b, dev)
        sd[B + "project.bias"] = _uniform(gen, (D,), b, dev)
    for w, (o, i) in dict(w1=(D, t.context_dim), w2=(D, D), w3=(D, t.context_dim)).items():
        lin(f"{P}y_embedder.projection.{w}", o, i)
    fd = t.frequency_embedding_dim
    for w, (o, i) in dict(w1=(D, fd), w2=(D, D), w3=(D, fd)).items():
        lin(f"{P}t_embedder.projection.{w}", o, i)
    lin(P + "t_block", 6 * D, D, bias=True)

    lin("proj", D, cfg.in_channels, bias=True)
    lin("memory_proj", D, cfg.text_encoder.dim, bias=True)
    vb = 1 / math.sqrt(cfg.vision_encoder.dim)
    sd["align_masked_video.conv.weight"] = _uniform(gen, (D, cfg.vision_encoder.dim, 1), vb, dev)
    sd["align_masked_video.conv.bias"] = _uniform(gen, (D,), 1.0, dev)
    norm_w("align_masked_video.layer_norm.weight", D)
    sd["align_masked_video.layer_norm.bias"] = _normal(gen, (D,), 0.1, dev)
    sd["align_masked_video.gate"] = torch.full((1,), 0.5, device=dev)
    emb = _normal(gen, (cfg.num_anchors + 1, cfg.anchor_embedding_dim), 1.0, dev)
    sd["embed_anchors.embed.weight"] = emb
    sd["embed_anchors.gate"] = torch.full((1,), 0.5, device=dev)
    lin("embed_anchors.proj", D, cfg.anchor_embedding_dim)

    if with_codec:
        sd.update(init_codec_state_dict(cfg, gen, dev))
    return sd


def make_hostile(sd: Dict[str, torch.Tensor], cfg: SAMAudioConfig, seed: int = 0, outliers: int = 4, outlier_gain: float = 300.0,
                 table_std: float = 3.0, codec_gain_std: float = 0.4) -> Dict[str, torch.Tensor]:
    """A copy of a synthetic checkpoint with the statistics TRAINED networks of this family show and seeded init does not
    (the parity claim of the 16-bit modes must survive them):

    * a few RESIDUAL-STREAM OUTLIER CHANNELS - `outliers` channels whose input projection is `outlier_gain` times stronger and
      which every layer keeps feeding (their `wo` / `w2` output rows x 30): the RMSNorm statistics and the 16-bit copies of the
      stream are then dominated by a handful of values in the hundreds;
    * adaLN SCALE / SHIFT / GATE TABLES OF O(`table_std`) instead of O(1 / sqrt(D)), and a `t_block` whose output is of that order
      too (reference transformer.py:350-352,462-471: the shipped init is the small one);
    * SNAKE ALPHAS SPREAD OVER TWO DECADES (10^U(-1, 1)) and DAC convolutions with weight-norm-like per-channel gains
      (log-normal, sigma `codec_gain_std`; reference codec.py:45-56 builds weight-normed convolutions).

    Test infrastructure of the precision claims (tests/test_hostile_gpu.py); the values stay finite in fp32 by construction."""
    g = torch.Generator().manual_seed(1000 + seed)
    t = cfg.transformer
    D = t.dim
    out = {k: v.clone() for k, v in sd.items()}
    dev = next(iter(out.values())).device

    def rnd(*shape, std=1.0):
        return (torch.randn(*shape, generator=g) * std).to(dev)

    ch = torch.randperm(D, generator=g)[:outliers].to(dev)
    out["proj.weight"][ch] *= outlier_gain
    out["proj.bias"][ch] *= outlier_gain
    P = "transformer."
    for i in range(t.n_layers):
        L = f"{P}layers.{i}."
        out[L + "attention.wo.weight"][ch] *= 30.0
        out[L + "feed_forward.w2.weight"][ch] *= 30.0
        out[L + "scale_shift_table"] = rnd(6, D, std=table_std)
    out[P + "final_layer_scale_shift_table"] = rnd(2, D, std=table_std)
    out[P + "t_block.weight"] = out[P + "t_block.weight"] * (table_std * 4.0)
    out[P + "t_block.bias"] = rnd(6 * D, std=table_std / 2)
    for k in list(out):
        if k.startswith("audio_codec."):
            if k.endswith(".alpha"):
                out[k] = (10.0 ** (torch.rand(out[k].shape, generator=g) * 2 - 1)).to(dev)
            elif k.endswith(".weight") and out[k].dim() == 3:
                # ConvTranspose1d stores [C_in, C_out, k]; a per-INPUT-channel gain there is as good a weight-norm stand-in
                gain = torch.exp(torch.randn(out[k].shape[0], generator=g) * codec_gain_std).to(dev)
                out[k] = out[k] * gain[:, None, None]
    return out


def codec_layout(cfg: SAMAudioConfig):
    """Channel plan of the DAC-VAE (HF `dac` topology, transformers/models/dac/modeling_dac.py
    :175-264,407-474; constructor arguments at reference config.py:11-37)."""
    c = cfg.audio_codec
    enc_ch = [c.encoder_dim * (2 ** i) for i in range(len(c.encoder_rates) + 1)]  # 64..1024
    dec_ch = [c.decoder_dim // (2 ** i) for i in range(len(c.decoder_rates) + 1)]  # 1536..96
    return enc_ch, dec_ch


def init_codec_state_dict(cfg: SAMAudioConfig, gen, dev) -> Dict[str, torch.Tensor]:
    c = cfg.audio_codec
    enc_ch, dec_ch = codec_layout(cfg)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, co, ci, k, gain=1.0):
        b = gain / math.sqrt(ci * k)
        sd[name + ".weight"] = _uniform(gen, (co, ci, k), b, dev)
        sd[name + ".bias"] = _uniform(gen, (co,), b, dev)

    def snake(name, ch):
        sd[name + ".alpha"] = (1.0 + _normal(gen, (1, ch, 1), 0.25, dev)).clamp(0.3, 2.0)

    def res_unit(name, ch):
        snake(name + ".block.0", ch)
        conv(name + ".block.1", ch, ch, 7)
        snake(name + ".block.2", ch)
        conv(name + ".block.3", ch, ch, 1, gain=0.5)

    E = "audio_codec.encoder.block."
    conv(E + "0", enc_ch[0], 1, 7, gain=2.0)
    for i, s in enumerate(c.encoder_rates):
        B = f"{E}{i + 1}.block."
        for j in range(3):
            res_unit(f"{B}{j}", enc_ch[i])
        snake(B + "3", enc_ch[i])
        conv(B + "4", enc_ch[i + 1], enc_ch[i], 2 * s)
    snake(E + "5", enc_ch[-1])
    conv(E + "6", c.latent_dim, enc_ch[-1], 3)
    conv("audio_codec.quantizer.in_proj", 2 * c.codebook_dim, c.latent_dim, 1, gain=2.0)
    conv("audio_codec.quantizer.out_proj", c.latent_dim, c.codebook_dim, 1)

    Dm = "audio_codec.decoder.model."
    conv(Dm + "0", dec_ch[0], c.latent_dim, 7)
    for i, s in enumerate(c.decoder_rates):
        B = f"{Dm}{i + 1}.block."
        snake(B + "0", dec_ch[i])
        b = 1 / math.sqrt(dec_ch[i] * 2)
        sd[B + "1.weight"] = _uniform(gen, (dec_ch[i], dec_ch[i + 1], 2 * s), b, dev)  # ConvTranspose1d
        sd[B + "1.bias"] = _uniform(gen, (dec_ch[i + 1],), b, dev)
        for j in range(3):
            res_unit(f"{B}{j + 2}", dec_ch[i + 1])
    snake(Dm + "5", dec_ch[-1])
    conv(Dm + "6", 1, dec_ch[-1], 7)
    return sd


def synthetic_clip(index: int, n_samples: int = 480_000, sample_rate: int = 48_000) -> torch.Tensor:
    """Clip `index` of the synthetic benchmark set: 0.1*noise + two sines,
    mono fp32 [1, n_samples] in [-1, 1]."""
    g = torch.Generator().manual_seed(1234 + index)
    t = torch.arange(n_samples, dtype=torch.float32) / sample_rate
    wav = 0.1 * torch.randn(n_samples, generator=g)
    wav += 0.2 * torch.sin(2 * math.pi * (220.0 + 10 * index) * t)
    wav += 0.2 * torch.sin(2 * math.pi * 1300.0 * t)
    return wav.clamp_(-1, 1).unsqueeze(0)


def synthetic_text_features(batch: int, text_len: int = 8, dim: int = 768, seed: int = 7,
                            ragged: bool = False):
    """Stand-in for the T5 encoder output (the t5-base tokenizer file is not available offline):
    (features [B, Lt, dim] fp32, mask [B, Lt] bool)."""
    g = torch.Generator().manual_seed(seed)
    feats = torch.randn(batch, text_len, dim, generator=g)
    mask = torch.ones(batch, text_len, dtype=torch.bool)
    if ragged:
        for b in range(batch):
            mask[b, max(1, text_len - (b % text_len)):] = False
    return feats, mask


def synthetic_noise(batch: int, frames: int, channels: int = 256, seed: int = 99) -> torch.Tensor:
    """ODE start state, generated on the CPU so CPU-oracle and GPU runs share it (quirk Q12)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn(batch, frames, channels, generator=g)


# ----------------------------------------------------------------------------------------------
# Judge reranker / PE-A-Frame span predictor
# ----------------------------------------------------------------------------------------------
def init_peav_state_dict(tc, prefix: str, gen, dev, bias_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Random weights of one PE-AV transformer.  Key names follow the Hugging Face port
    (transformers/models/pe_audio/modeling_pe_audio.py:241-287,344-490,616-640)."""
    D, Fh, hd = tc.hidden_size, tc.intermediate_size, tc.head_dim
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=False):
        b = 1.0 / math.sqrt(in_f)
        sd[prefix + name + ".weight"] = _uniform(gen, (out_f, in_f), b, dev)
        if bias:
            sd[prefix + name + ".bias"] = _uniform(gen, (out_f,), b * bias_gain, dev)

    def norm_w(name, n):
        sd[prefix + name] = 1.0 + _normal(gen, (n,), 0.1, dev)

    sd[prefix + "patch_embedder.class_embedding"] = _normal(gen, (1, 1, D), 0.5, dev)
    for blk in ("block1", "block2"):
        B = f"patch_embedder.resnet_block.{blk}."
        norm_w(B + "groupnorm.weight", D)
        sd[prefix + B + "groupnorm.bias"] = _normal(gen, (D,), 0.1, dev)
        b = 1 / math.sqrt(3 * D)
        sd[prefix + B + "project.weight"] = _uniform(gen, (D, D, 3), b, dev)
        sd[prefix + B + "project.bias"] = _uniform(gen, (D,), b, dev)
    for i in range(tc.num_hidden_layers):
        L = f"layers.{i}."
        for w in ("q_proj", "k_proj", "v_proj", "o_proj"):
            lin(L + "self_attn." + w, D, D, bias=tc.attention_bias)
        norm_w(L + "self_attn.q_norm.weight", hd)
        norm_w(L + "self_attn.k_norm.weight", hd)
        lin(L + "mlp.gate_proj", Fh, D)
        lin(L + "mlp.up_proj", Fh, D)
        lin(L + "mlp.down_proj", D, Fh)
        norm_w(L + "input_layernorm.weight", D)
        norm_w(L + "post_attention_layernorm.weight", D)
    norm_w("norm.weight", D)
    lin("output", D, D)
    return sd


def _encoder_only_codec(cfg_codec, gen, dev) -> Dict[str, torch.Tensor]:
    class _Wrap:  # init_codec_state_dict wants an object with .audio_codec
        audio_codec = cfg_codec
    full = init_codec_state_dict(_Wrap, gen, dev)
    return {k: v for k, v in full.items() if ".decoder." not in k and "quantizer.out_proj" not in k}


def init_judge_state_dict(cfg, seed: int = 0, device="cpu", with_codec: bool = True) -> Dict[str, torch.Tensor]:
    """Random checkpoint of the Judge (reference judge.py:39-74; the ModernBERT text tower is not part of it - the
    reference builds it with AutoModel.from_config and this build keeps it on PyTorch-ROCm).  `cfg` is a
    SAMAudioJudgeConfig.  mean/std are drawn non-trivially so that the de-normalisation is exercised."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    D, D2, Bn = cfg.transformer.hidden_size, cfg.finetune_transformer.hidden_size, cfg.bottleneck_dim
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f, bias=True):
        b = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = _uniform(gen, (out_f, in_f), b, dev)
        if bias:
            sd[name + ".bias"] = _uniform(gen, (out_f,), b, dev)

    lin("data_proj", D, cfg.audio_codec.codebook_dim)
    sd.update(init_peav_state_dict(cfg.transformer, "transformer.", gen, dev))
    sd.update(init_peav_state_dict(cfg.finetune_transformer, "finetune_transformer.", gen, dev))
    lin("cat_audio_proj", Bn, 2 * D)
    lin("text_proj1", D, cfg.text_hidden, bias=False)
    lin("text_proj2", Bn, D)
    sd["layer_norm.weight"] = 1.0 + _normal(gen, (Bn,), 0.1, dev)
    sd["layer_norm.bias"] = _normal(gen, (Bn,), 0.1, dev)
    lin("proj_audio_and_text", Bn, 2 * Bn)
    lin("finetune_data_proj", D2, Bn)
    lin("head", 4, D2, bias=False)
    sd["mean"] = _normal(gen, (4,), 1.0, dev)
    sd["std"] = 0.5 + torch.rand(4, generator=gen, device=dev)
    if with_codec:
        sd.update(_encoder_only_codec(cfg.audio_codec, gen, dev))
    return sd


def init_frame_state_dict(cfg, seed: int = 0, device="cpu") -> Dict[str, torch.Tensor]:
    """Random checkpoint of the PE-A-Frame span predictor's audio side + heads (`cfg` is a PEAudioFrameConfig;
    key names follow the HF port, modeling_pe_audio.py:158-195,721-735)."""
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    D, E = cfg.audio.hidden_size, cfg.text_hidden
    sd: Dict[str, torch.Tensor] = {}
    b = 1.0 / math.sqrt(cfg.codebook_dim)
    sd["audio_encoder.embedder.data_proj.weight"] = _uniform(gen, (D, cfg.codebook_dim), b, dev)
    sd["audio_encoder.embedder.data_proj.bias"] = _uniform(gen, (D,), b, dev)
    sd.update(init_peav_state_dict(cfg.audio, "audio_encoder.", gen, dev))
    for head, width in (("audio_head.", D), ("text_audio_head.", E)):
        sd[head + "layer_norm.weight"] = 1.0 + _normal(gen, (width,), 0.1, dev)
        sd[head + "layer_norm.bias"] = _normal(gen, (width,), 0.1, dev)
        sd[head + "proj.weight"] = _uniform(gen, (E, width), 1.0 / math.sqrt(width), dev)
    sd["text_audio_logit_scale"] = torch.full((1,), 4.0, device=dev)
    sd["text_audio_logit_bias"] = torch.full((1,), -0.5, device=dev)
    return sd


def init_vision_state_dict(cfg, seed: int = 0, device="cpu", prefix: str = "") -> Dict[str, torch.Tensor]:
    """Random weights of the PE-Core vision tower (config.PEVisionConfig) under the key names `pe.CLIP`'s vision tower
    uses below `visual.` (reference sam_audio/model/vision_encoder.py:86: `self.model = pe.CLIP.from_config(name)`, so a
    reference checkpoint carries them as `vision_encoder.model.visual.*`).  Scales: width**-0.5 for the class / position
    tables and `proj` (the published init), U(+-1/sqrt(fan_in)) for linear layers, LayerNorm weights around 1."""
    dev = torch.device(device)
    gen = None
    if dev.type != "meta":
        gen = torch.Generator(device=dev)
        gen.manual_seed(seed)
    W, Fw = cfg.width, cfg.mlp_width
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out_f, in_f):
        b = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = _uniform(gen, (out_f, in_f), b, dev)
        sd[name + ".bias"] = _uniform(gen, (out_f,), b, dev)

    def ln(name):
        sd[name + ".weight"] = 1.0 + _normal(gen, (W,), 0.1, dev)
        sd[name + ".bias"] = _normal(gen, (W,), 0.1, dev)

    def mha_block(name):
        b = 1.0 / math.sqrt(W)
        sd[name + ".in_proj_weight"] = _uniform(gen, (3 * W, W), b, dev)
        sd[name + ".in_proj_bias"] = _uniform(gen, (3 * W,), b, dev)
        lin(name + ".out_proj", W, W)

    kk = 3 * cfg.patch_size * cfg.patch_size
    sd["conv1.weight"] = _uniform(gen, (W, 3, cfg.patch_size, cfg.patch_size), 1.0 / math.sqrt(kk), dev)
    if cfg.use_cls_token:
        sd["class_embedding"] = _normal(gen, (W,), W ** -0.5, dev)
    if cfg.use_abs_posemb:
        sd["positional_embedding"] = _normal(gen, (cfg.tokens, W), W ** -0.5, dev)
    if cfg.use_ln_pre:
        ln("ln_pre")
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        ln(p + "ln_1")
        mha_block(p + "attn")
        ln(p + "ln_2")
        lin(p + "mlp.c_fc", Fw, W)
        lin(p + "mlp.c_proj", W, Fw)
    if cfg.use_ln_post:
        ln("ln_post")
    if cfg.pool_type == "attn":
        sd["attn_pool.probe"] = _normal(gen, (1, 1, W), 1.0, dev)
        mha_block("attn_pool.attn")
        ln("attn_pool.layernorm")
        lin("attn_pool.mlp.c_fc", Fw, W)
        lin("attn_pool.mlp.c_proj", W, Fw)
    sd["proj"] = _normal(gen, (W, cfg.output_dim), W ** -0.5, dev)
    return {prefix + k: v for k, v in sd.items()}
