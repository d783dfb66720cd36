#!/usr/bin/env python
"""Diagnostic for tests/test_path_gpu.py::test_concurrent_streams_are_bitwise_equal_to_one_stream: where does a two-stream
solve differ from the one-stream solve - by sharding (the same row groups solved one after the other on one stream), by
concurrency, or by a kernel form (debug flags)?  Prints max |diff| per clip for each variant, three repetitions."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config  # noqa: E402
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features  # noqa: E402

gpu = torch.device("cuda:0")
cfg = preset_config("mini")
sd = init_state_dict(cfg, seed=11)
hop = cfg.audio_codec.hop_length
clips = [synthetic_clip(i, 12 * hop) for i in range(5)]
text, tmask = synthetic_text_features(5, 6, ragged=True)
proc = SAMAudioProcessor.from_config(cfg)
batch = proc(descriptions=["x"] * 5, audios=clips, text_features=text, text_mask=tmask).to(gpu)
noise = synthetic_noise(5, 12).to(gpu)
opt = {"method": "midpoint", "options": {"step_size": 0.25}}


def model(streams, serial=False):
    m = SAMAudio(cfg, precision="bf16", device=str(gpu), streams=streams)
    m.load_state_dict(sd, strict=False)
    m._serial_groups = serial
    return m


def solve(m):
    m.separate(batch, noise=noise, ode_opt=opt)
    torch.cuda.synchronize()
    return m.last_latent.clone()


ref = solve(model(1))
for name, streams, serial, flags in (("one stream again", 1, False, {}), ("two groups, serial", 2, True, {}),
                                     ("two groups, concurrent", 2, False, {}), ("two groups, concurrent, flag 21", 2, False, {21: 1}),
                                     ("one stream, flag 21", 1, False, {21: 1})):
    for k, v in flags.items():
        hip.lib().samaudio_debug_set_flag(k, v)
    m = model(streams, serial)
    for rep in range(3):
        d = (solve(m).float() - ref.float()).abs().flatten(1).max(dim=1).values.tolist()
        print(f"{name:36s} rep {rep}: max|diff| per clip = {['%.3g' % x for x in d]}", flush=True)
    for k in flags:
        hip.lib().samaudio_debug_set_flag(k, 0)
