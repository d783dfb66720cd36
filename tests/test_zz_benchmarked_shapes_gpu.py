"""Parity at the BENCHMARKED shapes of BASELINE configs[3] and configs[4] - slow tests, outside bench.py's budget.

bench.py's own per-config checks run on reduced shapes so that the default invocation stays within minutes (configs[3]: 1 clip x 2
candidates of 1.28 s; configs[4]: 2.56 s clips).  Here the same comparisons run at the shapes the lines are timed on:

* configs[3]: 1 clip x 8 candidates x 10 s through DAC encode -> the 16-step solve of every candidate -> decode -> the Judge's
  scores -> argmax (reference model.py:297-330, model/judge.py:90-132, ranking/judge.py:21-42) against the oracles: candidate
  latents <= 1e-3, the same argmax (or a tie at the Judge's resolution: bench.parity_rerank), the selected waveform <= 1e-3,
  scores within the stated bound;
* configs[4]: 2 clips x 10 s with a full 250-frame masked video each through the PE-Core tower + the visually conditioned solve
  (reference model.py:186-191, vision_encoder.py:47-113): tower features vs the CPU tower oracle, latent and waveform <= 1e-3.

The DiT runs in the headline precision (fp16x3); the towers beside it on plain fp16 operands, as in bench.py.  Dims: `small*` by default
(the CPU oracle passes take ~3 minutes on the GPU box's 16 cores); SAMAUDIO_SHAPES_SIZE='large*' runs the benchmarked dims
(~25 minutes of oracle time: run by hand, recorded in profiles/r6_final/).
"""
import os

import pytest
import torch

import bench
from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
from sam_audio_amd.synthetic import init_state_dict

pytestmark = pytest.mark.gpu
SIZE = os.environ.get("SAMAUDIO_SHAPES_SIZE", "small*")
PREC = os.environ.get("SAMAUDIO_SHAPES_PRECISION", "fp16x3")
SIM = os.environ.get("SAMAUDIO_EMU_DRYRUN", "") != ""
# The 8-candidate test carries ~3 minutes of CPU oracle time (8 x 10 s rows + the Judge oracle on 16 waveforms): it runs when
# SAMAUDIO_SLOW_TESTS=1 (tools/r6_final.sh sets it: profiles/r6_final2/gpu_tests.log, shapes_large.log), so that the default
# `pytest -m gpu` stays near the duration of the earlier rounds' suites; the visual test (45 s) always runs.
SLOW = os.environ.get("SAMAUDIO_SLOW_TESTS", "") not in ("", "0")


def _model(gpu):
    cfg = preset_config(SIZE)
    sd = init_state_dict(cfg, seed=0, device=gpu)
    model = SAMAudio(cfg, precision=PREC, device=str(gpu))
    model.load_state_dict(sd, strict=False)
    return cfg, model, {k: v.cpu() for k, v in sd.items()}


@pytest.mark.skipif(SIM or not SLOW, reason="10 s clips x 8 candidates, ~3 min of CPU oracle: hardware, SAMAUDIO_SLOW_TESTS=1")
def test_eight_candidates_of_ten_seconds_through_spans_solve_and_judge(gpu):
    cfg, model, sd_cpu = _model(gpu)
    tower = bench.tower_precision(PREC)
    model.text_ranker, judge_sd = bench.build_judge_ranker(cfg, tower, gpu)
    judge_sd_cpu = {k: v.float().cpu() for k, v in judge_sd.items()}
    proc = SAMAudioProcessor.from_config(cfg)
    out = bench.parity_rerank(model, cfg, sd_cpu, proc, gpu, PREC, judge_sd_cpu, 10.0, 8, bench.usable_cores(), cand=8)
    print(f"configs[3] at its benchmarked shape ({SIZE}, {PREC} DiT, {tower} Judge): latent {out['ode_latent_err']:.3e} (|ref| <= "
          f"{out['ode_latent_ref_max']:.2f}), Judge scores {out['judge_score_err']:.3e}, argmax {out['argmax_hip']} / {out['argmax_oracle']} (margin by the oracle's scores {out['argmax_margin_oracle']:.2e}), "
          f"selected waveform {out['selected_waveform_err']}; oracle {out['oracle_seconds']}")
    assert out["rows"] == 8 and out["clip_seconds"] == 10.0
    assert out["ode_latent_err"] <= 1e-3
    # the same pick, or - by the oracle's own scores - a pick within twice the measured score error of the oracle's best (random-init
    # candidates score within 2e-4 of each other at large* dims: bench.parity_rerank); the waveform is the HIP pick's against the
    # oracle's waveform of that candidate
    assert out["argmax_consistent"], (out["argmax_margin_oracle"], out["judge_overall_scores_hip"], out["judge_overall_scores_oracle"])
    assert out["selected_waveform_err"] <= 1e-3
    # the Judge itself runs on plain 16-bit operands: 2 x the error measured on MI355X (profiles/r6_final/)
    assert out["judge_score_err"] <= 2e-3


@pytest.mark.skipif(SIM, reason="250 frames at 336 x 336 through the tower: hardware only")
def test_full_length_video_through_the_tower_and_the_visual_solve(gpu):
    from sam_audio_amd.config import PE_VISION_CONFIGS
    from sam_audio_amd.synthetic import init_vision_state_dict
    from sam_audio_amd.vision_encoder import PerceptionEncoder
    cfg, model, sd_cpu = _model(gpu)
    pe_cfg = PE_VISION_CONFIGS[cfg.vision_encoder.name]
    tower = bench.tower_precision(PREC)
    model.vision_encoder = PerceptionEncoder(cfg.vision_encoder, device=gpu, precision=tower)
    vsd = init_vision_state_dict(pe_cfg, seed=5, device=gpu)
    vsd_cpu = {k: v.float().cpu() for k, v in vsd.items()}
    model.vision_encoder.load_state_dict({"model.visual." + k: v for k, v in vsd.items()})
    proc = SAMAudioProcessor.from_config(cfg)
    out = bench.parity_visual(model, cfg, sd_cpu, proc, gpu, PREC, vsd_cpu, pe_cfg, 10.0, 8, bench.usable_cores())
    print(f"configs[4] at its benchmarked shape ({SIZE}, {PREC} DiT, {tower} tower, 250 frames per clip): tower features "
          f"{out['tower_feature_err']:.3e} (min cosine {out['tower_feature_min_cosine']:.7f}), latent {out['ode_latent_err']:.3e}, waveform "
          f"{out['waveform_err']:.3e}; oracle {out['oracle_seconds']}")
    assert out["clip_seconds"] == 10.0
    assert out["ode_latent_err"] <= 1e-3 and out["waveform_err"] <= 1e-3
    assert out["tower_feature_err"] <= 5e-4 and out["tower_feature_min_cosine"] >= 0.99999   # 16-bit tower: 2 x measured
