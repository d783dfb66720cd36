"""Parity at the configuration bench.py times: `large*` dims (D=2816, H=22, L=22, F=7552), T=250 frames, Lt=8, in
both precisions, against the CPU oracle (oracle/samaudio_oracle.py, pinned to the reference's own classes by
tests/test_oracle_golden.py).

north_star bound: 1e-3 max-abs.  fp32 mode is held to it.  The 16-bit operand modes are held to the bound stated in
BOUND below = (error measured on MI355X, printed by the test and quoted in README / DESIGN section 4) x 2, so that a
regression of the operand path fails here and not only in a listening test.  Weights are created on the GPU (2.9 B
parameters) and copied to the host for the oracle: ~12 GB of host memory, ~10 s of oracle time in total.
"""
import pytest
import torch

from oracle import samaudio_oracle as O
from sam_audio_amd import SAMAudio, preset_config
from sam_audio_amd.synthetic import init_state_dict, synthetic_noise
from tests import util

pytestmark = pytest.mark.gpu

# measured on MI355X with the default fp32 classes (out / in / prep GEMMs on exact-fp32 operands inside the 16-bit engines;
# profiles/r3_call1/gpu_tests_precision.log): forward fp32 5.5e-6, fp16 5.2e-4, bf16 4.2e-3 on |out| <= 2.7; 2-step
# midpoint latent fp32 3.8e-6, fp16 4.9e-4, bf16 3.4e-3 on |latent| <= 5.6 (round 2, every GEMM on 16-bit operands: bf16
# 8.8e-3 / 7.6e-3, fp16 1.31e-3 / 9.8e-4).  fp32 AND fp16 are held to the north_star's 1e-3 itself; bf16 = 2 x measured.
BOUND = {"fp32": 1e-3, "bf16": 9e-3, "fp16": 1e-3, "mixed": 1e-3}


@pytest.fixture(scope="module")
def large(gpu):
    cfg = preset_config("large*")
    sd = init_state_dict(cfg, seed=21, device=gpu, with_codec=False)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    B, T, Lt = 2, 250, 8
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, T, 128, generator=g)
    feats = torch.cat([z, z], 2)
    text = torch.randn(B, Lt, 768, generator=g)
    tmask = torch.ones(B, Lt, dtype=torch.bool)
    tmask[1, 5:] = False                                  # ragged text mask
    pad = torch.ones(B, T, dtype=torch.bool)
    pad[1, 200:] = False                                  # ragged clip length
    ids, align = O.anchors_to_ids([[("+", 1.0, 2.5)], []], pad, cfg.audio_codec.hop_length, cfg.audio_codec.sample_rate)
    noisy = synthetic_noise(B, T)
    time = torch.tensor([0.3125, 0.3125])
    video = torch.zeros(B, cfg.vision_encoder.dim, T)
    cond = dict(feats=feats, text=text, tmask=tmask, pad=pad, ids=ids, align=align, video=video)

    def field(t, y):
        return O.samaudio_forward(sd_cpu, cfg, y, feats, text, t.expand(B), video=video, text_mask=tmask,
                                  anchor_ids=ids, anchor_alignment=align, pad_mask=pad)

    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    with torch.inference_mode():
        want_fwd = field(time[:1], noisy)
        want_ode = O.ode_fixed_grid(field, noisy, method="midpoint", step_size=0.5)   # 2 steps = 4 evaluations
    return dict(cfg=cfg, sd=sd, cond=cond, noisy=noisy, time=time, want_fwd=want_fwd, want_ode=want_ode)


def _model(large, prec, gpu):
    m = SAMAudio(large["cfg"], precision=prec, device=str(gpu))
    m.load_state_dict(large["sd"], strict=False)
    return m


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "mixed"])
def test_forward_large_dims(gpu, large, prec):
    c = large["cond"]
    model = _model(large, prec, gpu)
    out = model.forward(large["noisy"], c["feats"], c["text"], large["time"], masked_video_features=c["video"],
                        text_mask=c["tmask"], anchor_ids=c["ids"], anchor_alignment=c["align"],
                        audio_pad_mask=c["pad"])
    util.report(f"large* forward {prec}", out, large["want_fwd"], BOUND[prec])


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16", "mixed"])
def test_two_step_midpoint_large_dims(gpu, large, prec):
    c = large["cond"]
    model = _model(large, prec, gpu)
    model._prepare(c["feats"], c["text"], c["tmask"], c["video"], c["ids"], c["align"], c["pad"])
    lat = model.solve(large["noisy"].to(gpu), {"method": "midpoint", "options": {"step_size": 0.5}})
    util.report(f"large* 2-step midpoint latent {prec}", lat, large["want_ode"], BOUND[prec])


# ---- the full default solve at the benchmarked dims -----------------------------------------
# separate() exactly as bench.py times it - DAC encode -> 16 midpoint steps = 32 DiT evaluations (reference model.py:22,
# 285-290) -> DAC decode of target + residual (model.py:291-295) - on 2 clips of 10 s with fixed CPU noise, against the
# fp32 CPU oracle (~100 s of oracle time on 16 cores).  north_star: 1e-3 max-abs.  fp32 and fp16 (with the fp32 classes of
# SAMAUDIO_OPT_F32_CLASSES, the default) are asserted at 1e-3 itself on latent AND waveform; bf16 cannot meet it (half an
# ulp at 1.0 is 3.9e-3) and is held to FULL_BOUND = 2 x measured as a regression guard, not as a parity claim.
# measured (profiles/r3_call1/gpu_tests_precision.log): fp32 1.9e-6 / 5.7e-7, fp16 4.2e-4 / 1.9e-4, bf16 3.4e-3 / 1.55e-3
FULL_BOUND = {"fp32": (1e-3, 1e-3), "fp16": (1e-3, 1e-3), "mixed": (1e-3, 1e-3), "bf16": (7e-3, 3.2e-3)}


@pytest.fixture(scope="module")
def full(gpu):
    from sam_audio_amd import SAMAudioProcessor
    from sam_audio_amd.synthetic import synthetic_clip, synthetic_text_features
    cfg = preset_config("large*")
    sd = init_state_dict(cfg, seed=0, device=gpu)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    R = 2
    n = 10 * cfg.audio_codec.sample_rate
    clips = [synthetic_clip(i, n) for i in range(R)]
    text, tmask = synthetic_text_features(R, 8, seed=7)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["sound"] * R, audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(R, n // cfg.audio_codec.hop_length)
    torch.set_num_threads(max(1, min(32, len(__import__("os").sched_getaffinity(0)))))
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd_cpu, cfg, batch.audios, batch.sizes.long(), text, tmask, noise)
    return dict(cfg=cfg, sd=sd, batch=batch, noise=noise, lat=lat_ref, wav=t_ref + r_ref)


@pytest.mark.parametrize("prec", ["fp32", "fp16", "mixed", "bf16"])
def test_full_solve_and_decode(gpu, full, prec):
    model = SAMAudio(full["cfg"], precision=prec, device=str(gpu))
    model.load_state_dict(full["sd"], strict=False)
    res = model.separate(full["batch"].to(gpu), noise=full["noise"].to(gpu))
    lat_err = (model.last_latent.cpu() - full["lat"]).abs().max().item()
    wav_err = max((a.cpu() - b).abs().max().item() for a, b in zip(res.target + res.residual, full["wav"]))
    print(f"large* full solve (16 midpoint steps) + decode, {prec}: latent max-abs err {lat_err:.3e} (|ref| <= "
          f"{full['lat'].abs().max().item():.2f}), waveform {wav_err:.3e} (|ref| <= "
          f"{max(w.abs().max().item() for w in full['wav']):.2f})")
    assert lat_err <= FULL_BOUND[prec][0], f"latent {lat_err} > {FULL_BOUND[prec][0]}"
    assert wav_err <= FULL_BOUND[prec][1], f"waveform {wav_err} > {FULL_BOUND[prec][1]}"
