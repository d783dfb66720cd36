#!/usr/bin/env python
"""Run-to-run determinism of the hot path's pieces on one stream: DAC-VAE encode, DAC-VAE decode, one DiT evaluation, the
whole separate(), each `--reps` times on identical inputs, compared bit for bit with the first result.  A kernel with a
scheduling-dependent race (a missing wait in a hand-counted DMA pipeline) shows up here as a rare mismatch; debug flags
(--flags 16=1 ...) bisect between kernel forms.

    python tools/stress_determinism.py [--reps 300] [--config mini] [--flags 16=1]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config  # noqa: E402
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=300)
    ap.add_argument("--config", default="mini")
    ap.add_argument("--clips", type=int, default=5)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--flags", nargs="*", default=[])
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--what", nargs="*", default=["encode", "decode", "forward", "separate"])
    args = ap.parse_args()
    gpu = torch.device("cuda:0")
    for fl in args.flags:
        k, v = fl.split("=")
        hip.lib(hip.operands_for(args.precision)).samaudio_debug_set_flag(int(k), int(v))
    cfg = preset_config(args.config)
    sd = init_state_dict(cfg, seed=11)
    hop = cfg.audio_codec.hop_length
    n, T = args.clips, args.frames
    clips = [synthetic_clip(i, T * hop) for i in range(n)]
    text, tmask = synthetic_text_features(n, 6, ragged=True)
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["x"] * n, audios=clips, text_features=text, text_mask=tmask).to(gpu)
    noise = synthetic_noise(n, T).to(gpu)
    opt = {"method": "midpoint", "options": {"step_size": 0.25}}
    m = SAMAudio(cfg, precision=args.precision, device=str(gpu), streams=1)
    m.load_state_dict(sd, strict=False)

    def run(name, fn):
        ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        for rep in range(args.reps):
            out = fn()
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                d = (out.float() - ref.float()).abs().flatten(1).max(dim=1).values.tolist()
                bad += 1
                if bad <= 5:
                    print(f"  {name}: repetition {rep} differs, max |diff| per item {['%.3g' % x for x in d]}", flush=True)
        print(f"{name}: {bad} of {args.reps} repetitions differ from the first (flags {args.flags}, {args.config}, {n} clips x {T} frames)",
              flush=True)

    z = m.encode_audio(batch.audios)
    lat = torch.randn(2 * n, T, z.shape[-1], device=gpu)
    feats = torch.cat([z, z], dim=2)
    if "encode" in args.what:
        run("codec encode", lambda: m.encode_audio(batch.audios))
    if "decode" in args.what:
        run("codec decode", lambda: m.decode_audio(lat))
    if "forward" in args.what:
        t = torch.full((n,), 0.3, device=gpu)
        run("DiT forward", lambda: m.forward(noise, feats, text.to(gpu), t, text_mask=tmask.to(gpu),
                                             audio_pad_mask=batch.audio_pad_mask))
    if "separate" in args.what:
        def sep():
            m.separate(batch, noise=noise, ode_opt=opt)
            return m.last_latent
        run("separate (latent)", sep)


if __name__ == "__main__":
    main()
