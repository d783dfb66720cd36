"""Parity of the PE-Core vision tower on the HIP library (sam_audio_amd/csrc/vit.hip, vit_kernels.hip; SURVEY.md
section 8 rows a4 / f3) against oracle/vit_oracle.py (pinned to HF CLIP's ViT and torch's MultiheadAttention for the
blocks they share, tests/test_vit_oracle_cpu.py).  Everything goes through the C ABI (`samaudio_vit_*`).

Tolerances: fp32 mode = exact-fp32 MFMA GEMMs + fp32 streaming kernels -> summation-order noise only (1e-4 on O(1)
values).  bf16 mode = bf16 GEMM operands (weights and activations rounded once per GEMM), fp32 accumulation / residual
stream / LayerNorm / softmax: bounds are 2x the errors measured on MI355X, printed by the tests.
"""
import dataclasses

import pytest
import torch

from oracle import vit_oracle as V
from sam_audio_amd.config import PE_VISION_CONFIGS, PerceptionEncoderConfig
from sam_audio_amd.synthetic import init_vision_state_dict
from sam_audio_amd.vision_encoder import PerceptionEncoder
from sam_audio_amd.vision_tower import PEVisionTower

pytestmark = pytest.mark.gpu


def _frames(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 3, size, size, generator=g).clamp(-1, 1)   # the range Normalize(0.5, 0.5) produces


def _run(cfg, precision, gpu, n=3, seed=11, normalize=True):
    sd = init_vision_state_dict(cfg, seed=seed)
    x = _frames(n, cfg.image_size, seed + 1)
    taps = {}
    with torch.inference_mode():
        want_tok = None
        feats_ref = V.vision_tower(sd, cfg, x, taps)
        want_tok = taps[f"layer{cfg.layers - 1}"] if cfg.layers else taps["embed"]
        want = torch.nn.functional.normalize(feats_ref, dim=-1) if normalize else feats_ref
    tower = PEVisionTower(cfg, precision=precision, device=str(gpu))
    tower.load_state_dict({"visual." + k: v for k, v in sd.items()})
    got, tok = tower.encode_image(x.to(gpu), normalize=normalize, return_tokens=True)
    return got.cpu(), want, tok.cpu(), want_tok


@pytest.mark.parametrize("name", ["pe-tiny", "pe-mini"])
@pytest.mark.parametrize("normalize", [True, False])
def test_tower_fp32_matches_oracle(gpu, name, normalize):
    cfg = PE_VISION_CONFIGS[name]
    got, want, tok, want_tok = _run(cfg, "fp32", gpu, normalize=normalize)
    e_tok = (tok - want_tok).abs().max().item() / want_tok.abs().max().item()
    e = (got - want).abs().max().item() / want.abs().max().item()
    print(f"vit {name} fp32 normalize={normalize}: tokens rel {e_tok:.2e}, features rel {e:.2e}")
    assert e_tok < 1e-4 and e < 1e-4


@pytest.mark.parametrize("change", [dict(pool_type="tok"), dict(pool_type="avg"), dict(use_rope2d=False),
                                    dict(act="quick_gelu"), dict(use_ln_pre=False), dict(use_abs_posemb=False),
                                    dict(use_cls_token=False, pool_type="avg"), dict(use_ln_post=False)])
def test_tower_structure_flags_fp32(gpu, change):
    """every structural switch of PEVisionConfig (pooling types, RoPE, activation, LayerNorms, class token, positions)"""
    cfg = dataclasses.replace(PE_VISION_CONFIGS["pe-tiny"], **change)
    got, want, tok, want_tok = _run(cfg, "fp32", gpu, n=2, seed=21)
    e_tok = (tok - want_tok).abs().max().item() / want_tok.abs().max().item()
    e = (got - want).abs().max().item()
    print(f"vit flags {change}: tokens rel {e_tok:.2e}, features abs {e:.2e}")
    assert e_tok < 1e-4 and e < 1e-4


def test_head_dim_128_blocks_fp32_and_bf16(gpu):
    """blocks with 128-wide heads (the width the pooling head of PE-Core-L uses) through the same kernels"""
    cfg = dataclasses.replace(PE_VISION_CONFIGS["pe-mini"], heads=2, attn_pooler_heads=4)
    got, want, tok, want_tok = _run(cfg, "fp32", gpu, n=2, seed=31)
    assert (tok - want_tok).abs().max().item() / want_tok.abs().max().item() < 1e-4
    assert (got - want).abs().max().item() < 1e-4
    got, want, _, _ = _run(cfg, "bf16", gpu, n=2, seed=31)
    assert (got - want).abs().max().item() < BF16_FEATURE_BOUND


# L2-normalised features (|f_i| <= 1): measured on MI355X (profiles/r2_call5/gpu_tests.log) pe-tiny 1.7e-3, pe-mini 1.6e-3,
# PE-Core-L14-336 6.8e-4 max-abs; residual-stream tokens 2.1e-3 .. 3.3e-3 relative.  Bounds = 2x the largest.
BF16_FEATURE_BOUND = 3.5e-3
BF16_TOKEN_BOUND = 7e-3


@pytest.mark.parametrize("name", ["pe-tiny", "pe-mini"])
def test_tower_bf16_matches_oracle(gpu, name):
    cfg = PE_VISION_CONFIGS[name]
    got, want, tok, want_tok = _run(cfg, "bf16", gpu)
    e_tok = (tok - want_tok).abs().max().item() / want_tok.abs().max().item()
    e = (got - want).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=-1).min().item()
    print(f"vit {name} bf16: tokens rel {e_tok:.2e}, normalised features abs {e:.2e}, min cosine {cos:.5f}")
    assert e < BF16_FEATURE_BOUND and cos > 0.9999 and e_tok < BF16_TOKEN_BOUND


def test_ragged_frame_counts_share_one_workspace(gpu):
    """the workspace is re-planned when the frame count changes (1, 5, 2 frames through one tower); rows are independent"""
    cfg = PE_VISION_CONFIGS["pe-tiny"]
    sd = init_vision_state_dict(cfg, seed=3)
    tower = PEVisionTower(cfg, precision="fp32", device=str(gpu))
    tower.load_state_dict(sd)                          # bare tower keys are accepted too
    x = _frames(5, cfg.image_size, 4)
    full = tower.encode_image(x.to(gpu), normalize=True).cpu()
    one = tower.encode_image(x[:1].to(gpu), normalize=True).cpu()
    two = tower.encode_image(x[3:].to(gpu), normalize=True).cpu()
    assert torch.equal(one, full[:1]) and torch.equal(two, full[3:])


def test_perception_encoder_uint8_video_chunked(gpu):
    """reference vision_encoder.py:47-113 end to end: uint8 frames -> Resize(bicubic, antialias) -> /255 -> Normalize ->
    tower in chunks of batch_size -> pad_sequence; two videos of different length."""
    pe = PE_VISION_CONFIGS["pe-tiny"]
    ecfg = PerceptionEncoderConfig(dim=pe.output_dim, batch_size=3, name="pe-tiny", image_size=pe.image_size)
    enc = PerceptionEncoder(ecfg, device=gpu, precision="fp32")
    sd = init_vision_state_dict(pe, seed=8)
    enc.load_state_dict({"model.visual." + k: v for k, v in sd.items()} | {"model.logit_scale": torch.ones(())}, strict=True)
    g = torch.Generator().manual_seed(9)
    videos = [torch.randint(0, 256, (7, 3, 80, 64), generator=g, dtype=torch.uint8),
              torch.randint(0, 256, (4, 3, 80, 64), generator=g, dtype=torch.uint8)]
    out = enc(videos).cpu()
    assert out.shape == (2, 7, pe.output_dim)
    assert float(out[1, 4:].abs().max()) == 0.0       # time padding
    for i, v in enumerate(videos):
        x = torch.nn.functional.interpolate(v.float(), size=(pe.image_size, pe.image_size), mode="bicubic",
                                            antialias=True, align_corners=False).round().clamp(0, 255)
        x = (x / 255.0 - 0.5) / 0.5
        want = V.encode_image(sd, pe, x, normalize=True)
        assert (out[i, : v.shape[0]] - want).abs().max().item() < 1e-4


def test_strict_load_reports_missing_tower_keys(gpu):
    cfg = PE_VISION_CONFIGS["pe-tiny"]
    sd = init_vision_state_dict(cfg, seed=1)
    sd.pop("attn_pool.probe")
    tower = PEVisionTower(cfg, precision="fp32", device=str(gpu))
    with pytest.raises(RuntimeError, match="attn_pool.probe"):
        tower.load_state_dict(sd)
    with pytest.raises(Exception):
        tower.encode_image(_frames(1, cfg.image_size, 1).to(gpu))


@pytest.mark.skipif("SAMAUDIO_EMU_DRYRUN" in __import__("os").environ, reason="full-size tower: MI355X only")
@pytest.mark.parametrize("precision,bound", [("fp32", 2e-4), ("bf16", BF16_FEATURE_BOUND)])
def test_pe_core_l14_336_dims(gpu, precision, bound):
    """The tower BASELINE configs[4] runs (PE-Core-L14-336: 577 tokens, width 1024, 24 layers, 16 x 64 heads, attention
    pooling with 8 x 128 heads), 2 frames, seeded random weights, vs the CPU oracle."""
    cfg = PE_VISION_CONFIGS["PE-Core-L14-336"]
    got, want, tok, want_tok = _run(cfg, precision, gpu, n=2, seed=41)
    e_tok = (tok - want_tok).abs().max().item() / want_tok.abs().max().item()
    e = (got - want).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=-1).min().item()
    print(f"PE-Core-L14-336 {precision}: tokens rel {e_tok:.2e}, normalised features abs {e:.2e}, min cosine {cos:.6f}")
    assert e < bound and cos > 0.9999 and e_tok < (BF16_TOKEN_BOUND if precision == "bf16" else 1e-4)


def test_separate_with_the_hip_tower_from_a_checkpoint(gpu):
    """Rows a4 + f3 end to end: a SAMAudio state_dict that carries the PE tower under `vision_encoder.model.visual.*`
    (reference model.py:82-83) builds the PerceptionEncoder + HIP tower on load; separate() with masked videos then
    runs processor frame sampling -> resize / normalise -> HIP tower -> the video term of align_inputs -> ODE, and the
    latent must match the oracle fed with the ORACLE tower's features."""
    import dataclasses as dc
    from oracle import samaudio_oracle as O
    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features
    from tests import util
    pe = PE_VISION_CONFIGS["pe-tiny"]
    cfg = preset_config("tiny")
    cfg.vision_encoder = PerceptionEncoderConfig(dim=pe.output_dim, batch_size=3, name="pe-tiny", image_size=pe.image_size)
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 4 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 3)
    g = torch.Generator().manual_seed(12)
    videos = [torch.randint(0, 256, (7, 3, 70, 60), generator=g, dtype=torch.uint8),
              torch.randint(0, 256, (5, 3, 56, 56), generator=g, dtype=torch.uint8)]
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["a", "b"], audios=clips, masked_videos=videos, text_features=text, text_mask=tmask)
    sd = init_state_dict(cfg, seed=3)
    vsd = init_vision_state_dict(pe, seed=6)
    full = dict(sd)
    full.update({"vision_encoder.model.visual." + k: v for k, v in vsd.items()})
    full["vision_encoder.model.logit_scale"] = torch.ones(())       # the CLIP pair's text side: present, not on this path
    noise = synthetic_noise(2, 4)

    def ref_features(v):
        x = v.float()
        if x.shape[-2:] != (pe.image_size, pe.image_size):
            x = torch.nn.functional.interpolate(x, size=(pe.image_size, pe.image_size), mode="bicubic", antialias=True,
                                                align_corners=False).round().clamp(0, 255)
        return V.encode_image(vsd, pe, (x / 255.0 - 0.5) / 0.5, normalize=True)

    with torch.inference_mode():
        feats_v = torch.stack([ref_features(v) for v in batch.masked_video])         # [B, T, dim]
        _, _, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise,
                                   video=feats_v.transpose(1, 2), decode=False)
        _, _, lat_novid = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise, decode=False)
    assert (lat_ref - lat_novid).abs().max() > 1e-3, "the video term must matter for this check to mean anything"
    model = SAMAudio(cfg, precision="fp32", device=str(gpu))
    assert model.vision_encoder is None
    # a genuine pe.CLIP state_dict may carry tensors under `visual.` that the restated key list does not know (buffers,
    # layer-scale): a strict load must report them, not fail on them
    extra = dict(full)
    extra["vision_encoder.model.visual.some_buffer_the_engine_does_not_use"] = torch.zeros(3)
    with pytest.warns(UserWarning, match="not consumed by the PE-Core tower"):
        model.load_state_dict(extra, strict=True)
    model.load_state_dict(full, strict=True)
    assert model.vision_encoder is not None and model.vision_encoder.tower.__class__.__name__ == "PEVisionTower"
    model.separate(batch.to(gpu), noise=noise.to(gpu))
    util.report("latent with the HIP tower's visual prompt", model.last_latent, lat_ref, 1e-3)


def test_two_streams_are_bitwise_equal_to_one(gpu):
    """`PEVisionTower(streams=2)` encodes batches of >= 64 frames as two halves on two HIP streams / engine contexts that
    share the weight tensors; frames are independent, so the features must equal the single-stream ones bit for bit."""
    if gpu.type != "cuda":
        pytest.skip("needs real HIP streams")
    cfg = PE_VISION_CONFIGS["pe-tiny"]
    sd = init_vision_state_dict(cfg, seed=13)
    x = _frames(70, cfg.image_size, 14).to(gpu)
    outs = []
    for streams in (1, 2):
        tower = PEVisionTower(cfg, precision="bf16", device=str(gpu), streams=streams)
        tower.load_state_dict(sd)
        for _ in range(2):
            outs.append(tower.encode_image(x, normalize=True).cpu())
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
