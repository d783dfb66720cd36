"""The 16-bit modes on TRAINED-LIKE statistics ("parity first" item 1).

Every other parity test of this suite runs on seeded random weights, whose activations are benign: unit-scale residual stream,
adaLN tables of O(1 / sqrt(D)), Snake alphas near 1.  `synthetic.make_hostile` plants what trained DiTs / codecs show - a few
residual-stream outlier channels in the hundreds, adaLN scale / shift / gate tables of O(3), a strong t_block, Snake alphas over two
decades, weight-norm-like gains on the DAC convolutions - and this file runs separate() as bench.py times it (DAC encode -> 16
midpoint steps = 32 DiT evaluations -> decode) on it, in every precision, against the fp32 CPU oracle, with the sentinel on
(samaudio.h SAMAUDIO_OPT_SENTINEL): an fp16 operand that overflowed is reported with its GEMM class.

Dims: `small*` by default (the oracle pass of 2 clips x 10 s is ~40 s on the GPU box's 16 cores); SAMAUDIO_HOSTILE_SIZE=large* runs the
benchmarked dims (tools: one run is recorded in DESIGN.md section 4).  fp32 mode is held to the north_star's 1e-3 - the HIP path
itself computes the hostile network correctly; the 16-bit modes are held to BOUND = 2 x the error measured on MI355X (a regression
guard; what the numbers mean for the choice of the headline mode is DESIGN.md section 4's subject), and NO mode may produce a
non-finite value anywhere.
"""
import os

import pytest
import torch

from oracle import samaudio_oracle as O
from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config
from sam_audio_amd.synthetic import init_state_dict, make_hostile, synthetic_clip, synthetic_noise, synthetic_text_features

pytestmark = pytest.mark.gpu
SIZE = os.environ.get("SAMAUDIO_HOSTILE_SIZE", "small*")
# (latent, waveform) max-abs bounds: fp32 = the north_star's; 16-bit = 2 x measured on MI355X (profiles/r5_call6/: small* fp16 3.2e-2 /
# 3.4e-3, mixed 3.5e-1 / 2.8e-2, bf16 2.1e-1 / 2.2e-2 on |latent| <= 14.0, |wave| <= 0.85; large* fp16 6.6e-3 / 2.7e-3, mixed 6.9e-2 /
# 1.7e-2, bf16 6.9e-2 / 2.7e-2 on |latent| <= 13.2, |wave| <= 0.94; fp32 8.9e-5 / 8.0e-6 and 1.4e-5 / 6.9e-6).  Reading: NO 16-bit mode
# holds 1e-3 on these statistics; IEEE fp16 operands are ten times closer than any mode with bfloat16 operands.
# Round 6: precision "fp16x3" (fp32 storage, every big contraction on hi/lo-split IEEE-half operands: 3 MFMA products per multiply,
# include/samaudio.h SAMAUDIO_OPT_X3_CLASSES) is held to the north_star's (1e-3, 1e-3) like fp32 - measured 9.8e-5 / 1.4e-5 at small*,
# 7.0e-5 / 1.3e-5 at large* (profiles/r6_call2/): the fast mode that holds the bound on these statistics, and bench.py's headline.
BOUNDS = {"small*": {"fp32": (1e-3, 1e-3), "fp16x3": (1e-3, 1e-3), "fp16": (6.5e-2, 7e-3), "mixed": (7e-1, 6e-2), "bf16": (4.5e-1, 4.5e-2)},
          "large*": {"fp32": (1e-3, 1e-3), "fp16x3": (1e-3, 1e-3), "fp16": (1.4e-2, 5.5e-3), "mixed": (1.4e-1, 3.5e-2), "bf16": (1.4e-1, 5.5e-2)}}
BOUND = BOUNDS.get(SIZE, {"fp32": (1e-3, 1e-3), "fp16x3": (1e-3, 1e-3), "fp16": (None, None), "mixed": (None, None), "bf16": (None, None)})
FP16_MAX = 65504.0


@pytest.fixture(scope="module")
def hostile(gpu):
    cfg = preset_config(SIZE)
    sd = make_hostile(init_state_dict(cfg, seed=0, device=gpu), cfg, seed=0)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    R = 2
    n = (10 if gpu.type == "cuda" else 1) * cfg.audio_codec.sample_rate // cfg.audio_codec.hop_length * cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, n) for i in range(R)]
    text, tmask = synthetic_text_features(R, 8, seed=7)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["sound"] * R, audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(R, n // cfg.audio_codec.hop_length)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd_cpu, cfg, batch.audios, batch.sizes.long(), text, tmask, noise)
    assert torch.isfinite(lat_ref).all()
    return dict(cfg=cfg, sd=sd, batch=batch, noise=noise, lat=lat_ref, wav=t_ref + r_ref)


@pytest.mark.parametrize("prec", ["fp32", "fp16x3", "fp16", "mixed", "bf16"])
def test_full_solve_on_hostile_weights(gpu, hostile, prec):
    model = SAMAudio(hostile["cfg"], precision=prec, device=str(gpu))
    model.load_state_dict(hostile["sd"], strict=False)
    model.sentinel(prec != "fp16x3")   # (an x3 context has no 16-bit activation tensors to scan: its operands are split on the fly)
    res = model.separate(hostile["batch"].to(gpu), noise=hostile["noise"].to(gpu))
    rep = model.sentinel_report()
    lat_ref, wav_ref = hostile["lat"], hostile["wav"]
    lat_err = (model.last_latent.cpu() - lat_ref).abs().max().item()
    wav_err = max((a.cpu() - b).abs().max().item() for a, b in zip(res.target + res.residual, wav_ref))
    lat_max, wav_max = lat_ref.abs().max().item(), max(w.abs().max().item() for w in wav_ref)
    seen = {k: v for k, v in rep.items() if v["absmax"] > 0 or v["nonfinite"] > 0}
    print(f"hostile {SIZE} full solve + decode, {prec}: latent max-abs err {lat_err:.3e} (|ref| <= {lat_max:.2f}), waveform "
          f"{wav_err:.3e} (|ref| <= {wav_max:.2f}); sentinel absmax per class: "
          + ", ".join(f"{k} {v['absmax']:.4g}" for k, v in seen.items()))
    assert torch.isfinite(model.last_latent).all() and all(torch.isfinite(w).all() for w in res.target + res.residual)
    bad = {k: v["nonfinite"] for k, v in rep.items() if v["nonfinite"] > 0}
    assert not bad, f"non-finite 16-bit tensors written by: {bad}"
    assert seen or prec == "fp16x3", "the sentinel saw no tensor: is it wired to the launches?"
    if prec in ("fp16", "mixed"):   # IEEE fp16 tensors: a factor 4 of headroom below the format's largest value
        alt = {c for c in hip.CLASSES if model.alt16_classes & hip.CLS[c]}   # these write / read bfloat16 in the mixed mode
        if prec == "mixed":   # ... and so do the RMSNorm / attention outputs that feed them (the 'norm' / 'attn' slots)
            alt |= {"norm", "attn"}
        close = {k: v["absmax"] for k, v in rep.items() if k not in alt and v["absmax"] > FP16_MAX / 4}
        assert not close, f"fp16 tensors within a factor 4 of overflow: {close}"
    b_lat, b_wav = BOUND[prec]
    if b_lat is not None:
        assert lat_err <= b_lat, f"latent {lat_err} > {b_lat}"
        assert wav_err <= b_wav, f"waveform {wav_err} > {b_wav}"
