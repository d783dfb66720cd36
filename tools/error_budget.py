#!/usr/bin/env python
"""Error budget of the 16-bit operand modes at the benchmarked dims (run on the GPU box).

    python tools/error_budget.py [--size 'large*'] [--clips 2] [--steps 16] [--out gpurun_out/error_budget.json]

The reference computes in fp32 (README.md:48).  This tool answers "which GEMM classes carry the error of the bf16 / fp16
modes" on the FULL solve (DAC encode -> 16 midpoint steps = 32 evaluations -> DAC decode):

1. baseline = the fp32 engine (exact-fp32 MFMA; 5e-6 from the CPU oracle, tests/test_large_gpu.py);
2. per class c and format f: the fp32 engine with ONLY the GEMMs of class c rounding both operands to f
   (SAMAudio.set_quantised_classes -> SAMAUDIO_OPT_QUANT_CLASSES) -> max-abs / rms difference of latent and waveform;
3. every class at once (the emulation of a whole 16-bit mode, minus the attention kernels' own operand roundings), and
   every class except the ones SAMAUDIO_OPT_F32_CLASSES can keep exact;
4. the real 16-bit engines with f32_classes = none / auto.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config  # noqa: E402
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_text_features  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="large*")
    ap.add_argument("--clips", type=int, default=2)
    ap.add_argument("--steps", type=int, default=16, help="midpoint steps of the solve (16 = the reference's default)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "error_budget.json"))
    ap.add_argument("--formats", default="bf16,fp16")
    ap.add_argument("--hostile", action="store_true",
                    help="trained-like statistics (synthetic.make_hostile: residual-stream outlier channels, O(3) adaLN tables, Snake "
                         "alphas over two decades, gains on the DAC convolutions) instead of the benign seeded init")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = preset_config(args.size)
    sd = init_state_dict(cfg, seed=0, device=dev)
    if args.hostile:
        from sam_audio_amd.synthetic import make_hostile
        sd = make_hostile(sd, cfg, seed=0)
    R = args.clips
    n_samples = 10 * cfg.audio_codec.sample_rate
    clips = [synthetic_clip(i, n_samples) for i in range(R)]
    text, tmask = synthetic_text_features(R, 8, seed=7)
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["sound"] * R, audios=clips, text_features=text, text_mask=tmask).to(dev)
    g = torch.Generator().manual_seed(99)
    noise = torch.randn(R, n_samples // cfg.audio_codec.hop_length, cfg.transformer.out_channels, generator=g).to(dev)
    ode = {"method": "midpoint", "options": {"step_size": 1.0 / args.steps}}

    def run(model):
        with torch.inference_mode():
            t0 = time.perf_counter()
            res = model.separate(batch, noise=noise, ode_opt=ode)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        wav = torch.stack([torch.stack(res.target), torch.stack(res.residual)], 1)
        return model.last_latent.clone(), wav.clone(), dt

    def diff(a, b):
        d = (a - b).float()
        return float(d.abs().max()), float(d.pow(2).mean().sqrt())

    out = {"size": args.size, "clips": R, "midpoint_steps": args.steps, "hostile": bool(args.hostile), "rows": []}
    m32 = SAMAudio(cfg, precision="fp32", device=str(dev))
    m32.load_state_dict(sd, strict=False)
    lat0, wav0, dt = run(m32)
    out["baseline"] = {"latent_max": float(lat0.abs().max()), "latent_rms": float(lat0.pow(2).mean().sqrt()),
                       "wave_max": float(wav0.abs().max()), "seconds": dt}
    print(f"fp32 baseline: |latent| <= {out['baseline']['latent_max']:.3f} (rms {out['baseline']['latent_rms']:.3f}), "
          f"|wave| <= {out['baseline']['wave_max']:.3f}, {dt:.2f} s", flush=True)

    def record(label, fmt, lat, wav, dt, kind):
        lm, lr = diff(lat, lat0)
        wm, wr = diff(wav, wav0)
        out["rows"].append({"what": label, "format": fmt, "kind": kind, "latent_maxabs": lm, "latent_rms": lr,
                            "wave_maxabs": wm, "wave_rms": wr, "seconds": dt})
        print(f"{kind:9s} {fmt:5s} {label:34s} latent max {lm:.3e} rms {lr:.3e} | wave max {wm:.3e} rms {wr:.3e} | {dt:.2f} s",
              flush=True)

    capable = [c for c in hip.CLASSES if hip.CLS[c] & hip.CLS_F32_CAPABLE]
    rest = [c for c in hip.CLASSES if c not in capable]
    for fmt in args.formats.split(","):
        for c in hip.CLASSES:
            m32.set_quantised_classes([c], fmt)
            record(c, fmt, *run(m32), "emulated")
        m32.set_quantised_classes(["all"], fmt)
        record("all classes", fmt, *run(m32), "emulated")
        m32.set_quantised_classes(rest, fmt)
        record("all but " + "+".join(capable), fmt, *run(m32), "emulated")
        m32.set_quantised_classes([c for c in rest if c != "codec"], fmt)
        record("DiT layer classes only (no codec)", fmt, *run(m32), "emulated")
    m32.set_quantised_classes([], "bf16")
    del m32
    torch.cuda.empty_cache()
    for fmt in args.formats.split(","):
        for label, classes in (("f32_classes = none", 0), ("f32_classes = auto", "auto"),
                               ("f32_classes = all five", "time,out,in,prep,yemb"), ("f32_classes = out+in", "out,in")):
            m = SAMAudio(cfg, precision=fmt, device=str(dev), f32_classes=classes)
            m.load_state_dict(sd, strict=False)
            run(m)   # warm-up (workspace allocation)
            record(label, fmt, *run(m), "real")
            del m
            torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
