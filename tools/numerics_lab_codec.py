#!/usr/bin/env python
"""CPU numerics lab for the DAC-VAE (TEST INFRASTRUCTURE, no GPU): which convolutions of the codec carry the fp16-operand error on the
hostile weights?  Patches F.conv1d / F.conv_transpose1d of the oracle per layer group with an operand-rounding emulation.

    python tools/numerics_lab_codec.py --seconds 2.56 --schemes fp16,x3
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import samaudio_oracle as O  # noqa: E402
from sam_audio_amd import preset_config  # noqa: E402
from sam_audio_amd.synthetic import init_state_dict, make_hostile, synthetic_clip  # noqa: E402


def h(x):
    return x.half().float()


def group_of(key):
    # audio_codec.decoder.model.{i}.block.{j}...   /  audio_codec.encoder.block.{i}.block.{j}...
    p = key.split(".")
    if "decoder" in p:
        i = int(p[3])
        if i == 0:
            return "dec.in"
        if i == 6:
            return "dec.out"
        j = int(p[5])
        return f"dec.s{i - 1}.up" if j == 1 else f"dec.s{i - 1}.res"
    if "encoder" in p:
        i = int(p[3])
        if i == 0:
            return "enc.in"
        if i == 6:
            return "enc.out"
        j = int(p[5])
        return f"enc.s{i - 1}.down" if j == 4 else f"enc.s{i - 1}.res"
    return "proj"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.56)
    ap.add_argument("--schemes", default="fp16")
    ap.add_argument("--benign", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(len(os.sched_getaffinity(0)))
    cfg = preset_config("small*")
    sd = init_state_dict(cfg, seed=0, device=torch.device("cpu"))
    if not args.benign:
        sd = make_hostile(sd, cfg, seed=0)
    groups = {id(v): group_of(k) for k, v in sd.items() if k.startswith("audio_codec.") and k.endswith(".weight")}
    names = sorted(set(groups.values()))
    hop = cfg.audio_codec.hop_length
    n = int(args.seconds * cfg.audio_codec.sample_rate) // hop * hop
    wav = torch.stack([synthetic_clip(i, n) for i in range(2)])
    real_c, real_t = F.conv1d, F.conv_transpose1d
    state = {"on": set(), "scheme": "fp32"}

    def q(fn, x, w, b, **kw):
        g = groups.get(id(w))
        if g not in state["on"] or state["scheme"] == "fp32":
            return fn(x, w, b, **kw)
        xh, wh = h(x), h(w)
        if state["scheme"] == "fp16":
            return fn(xh, wh, b, **kw)
        xl, wl = h(x - xh), h(w - wh)   # x3
        return fn(xh, wh, b, **kw) + fn(xl, wh, None, **kw) + fn(xh, wl, None, **kw)

    O.F.conv1d = lambda x, w, b=None, **kw: q(real_c, x, w, b, **kw)
    O.F.conv_transpose1d = lambda x, w, b=None, **kw: q(real_t, x, w, b, **kw)

    def run():
        with torch.inference_mode():
            z = O.dac_encode(sd, cfg.audio_codec, wav)
            zz = torch.cat([z, 0.5 * z.flip(0)], 0)   # a second pair of latents to decode
            state_keep = set(state["on"])
            return z, O.dac_decode(sd, cfg.audio_codec, zz)

    t0 = time.time()
    z0, w0 = run()
    print(f"{'benign' if args.benign else 'hostile'} codec, {args.seconds} s clips: |latent| <= {z0.abs().max():.3f}, |wave| <= {w0.abs().max():.3f} "
          f"({time.time() - t0:.1f} s)", flush=True)
    for scheme in args.schemes.split(","):
        state["scheme"] = scheme
        for sel in [[n_] for n_ in names] + [names, [n_ for n_ in names if n_.startswith("enc")], [n_ for n_ in names if n_.startswith("dec")]]:
            state["on"] = set(sel)
            z, w = run()
            label = sel[0] if len(sel) == 1 else ("all" if sel == names else sel[0][:3] + ".*")
            print(f"{scheme:5s} {label:12s} latent max-abs {(z - z0).abs().max():.3e}   wave max-abs {(w - w0).abs().max():.3e}", flush=True)


if __name__ == "__main__":
    main()
