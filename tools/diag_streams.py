"""Diagnostic for tests/test_path_gpu.py::test_concurrent_streams_are_bitwise_equal_to_one_stream."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features
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
def mk(streams):
    m = SAMAudio(cfg, precision="bf16", device=str(gpu), streams=streams)
    m.load_state_dict(sd, strict=False)
    return m
from sam_audio_amd import hip
for flags in ((), ((14, 1),), ((15, 1),), ((14, 1), (15, 1))):
    for k in (14, 15):
        hip.lib().samaudio_debug_set_flag(k, 0)
    for k, v in flags:
        hip.lib().samaudio_debug_set_flag(k, v)
    one = mk(1)
    one.separate(batch, noise=noise, ode_opt=opt)
    ref = one.last_latent.clone()
    f0 = one.encode_audio(batch.audios).clone()
    one.separate(batch, noise=noise, ode_opt=opt)
    f1 = one.encode_audio(batch.audios)
    print("flags", flags, "one-stream repeat: latent diff", (one.last_latent - ref).abs().max().item(),
          "encode repeat diff", (f1 - f0).abs().max().item())
for k in (14, 15):
    hip.lib().samaudio_debug_set_flag(k, 0)
sys.exit(0)
for split in (None, True, False):
    two = mk(2)
    two.tail_split = split
    for it in range(3):
        two.separate(batch, noise=noise, ode_opt=opt)
        torch.cuda.synchronize()
        d = (two.last_latent - ref).abs().amax(dim=(1, 2)).tolist()
        print(f"two streams tail_split={split} iter {it}: per-clip max diff", [f"{x:.2e}" for x in d])
# decode off: patch
import types
two = mk(2)
orig = two._solve_concurrent
two._solve_concurrent = lambda noise, ode_opt, cond, groups, decode=False: (orig(noise, ode_opt, cond, groups, decode=False), None)
for it in range(2):
    two.separate(batch, noise=noise, ode_opt=opt)
    torch.cuda.synchronize()
    d = (two.last_latent - ref).abs().amax(dim=(1, 2)).tolist()
    print(f"two streams, decode on the main stream afterwards, iter {it}:", [f"{x:.2e}" for x in d])
# shards through ONE context, one stream
m = mk(1)
for sl in (slice(0, 3), slice(3, 5)):
    b2 = proc(descriptions=["x"] * (sl.stop - sl.start), audios=clips[sl], text_features=text[sl], text_mask=tmask[sl]).to(gpu)
    m.separate(b2, noise=noise[sl], ode_opt=opt)
    print("shard", sl, "diff", (m.last_latent - ref[sl]).abs().max().item())
