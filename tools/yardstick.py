#!/usr/bin/env python
"""Same-box yardsticks (test infrastructure, never the product; run on the GPU box):

  gemm   - hipBLASLt (`torch.matmul` on bf16 tensors) on the five DiT GEMM shapes at M = 8000 / 4000 / 1000 rows, beside this
           build's kernels on the same shapes through the C ABI (plain epilogue: 16-bit output, like the library call);
  eager  - the oracle modules (oracle/samaudio_oracle.py = the reference's op sequence in plain torch) moved onto the GPU:
           "the reference on this GPU through PyTorch-ROCm".  One midpoint step (2 of the 32 DiT evaluations) of BASELINE
           configs[2] (32 clips, large* dims) is timed in fp32 and under bf16 autocast and scaled x16; the DAC-VAE encode /
           decode are timed once on 4 waveforms and scaled.

    python tools/yardstick.py gemm|eager|all [--out gpurun_out/yardstick.json]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sam_audio_amd import hip, preset_config  # noqa: E402


def ev_time(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def gemm_yardstick(size="large*"):
    from tests import util
    t = preset_config(size).transformer
    D, F = t.dim, t.ffn_hidden
    shapes = [("qkv", 3 * D, D), ("c_wq", D, D), ("wo", D, D), ("w13", 2 * F, D), ("w2", D, F)]
    dev = torch.device("cuda:0")
    rows = []
    for M in (8000, 4000, 1000):
        for name, N, K in shapes:
            g = torch.Generator(device=dev).manual_seed(1)
            A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
            W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
            flops = 2.0 * M * N * K
            t_blas = ev_time(lambda: torch.matmul(A, W.t()))
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            t_ours = ev_time(lambda: util.gemm("bf16", A, W, M, N, K, out_act=out, act_geom=(0, N, 0)))
            err = (out.float() - torch.matmul(A, W.t()).float()).abs().max().item()
            rows.append({"gemm": name, "M": M, "N": N, "K": K, "hipblaslt_tflops": round(flops / t_blas / 1e12, 1),
                         "ours_tflops": round(flops / t_ours / 1e12, 1), "ratio": round(t_blas / t_ours, 3),
                         "max_abs_diff": err})
            print(f"M={M:5d} {name:5s} N={N:6d} K={K:5d}: hipBLASLt {rows[-1]['hipblaslt_tflops']:7.1f} TF/s | "
                  f"ours {rows[-1]['ours_tflops']:7.1f} TF/s | x{rows[-1]['ratio']:.2f} | diff {err:.2e}", flush=True)
    return rows


def eager_yardstick(size="large*", clips=32):
    from oracle import samaudio_oracle as O
    from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_text_features
    dev = torch.device("cuda:0")
    cfg = preset_config(size)
    sd = init_state_dict(cfg, seed=0, device=dev)
    B, T = clips, 250
    # inputs are drawn on the CPU (seeded CPU generators), then moved; only then do factory calls default to the GPU
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, T, 128, generator=g).to(dev)
    noise = torch.randn(B, T, 256, generator=g).to(dev)
    feats = torch.cat([z, z], 2)
    text, tmask = synthetic_text_features(B, 8, seed=7)
    text, tmask = text.to(dev), tmask.to(dev)
    ids, align = O.anchors_to_ids(None, torch.ones(B, T, dtype=torch.bool), cfg.audio_codec.hop_length,
                                  cfg.audio_codec.sample_rate)
    ids, align = ids.to(dev), align.to(dev)
    n = 10 * cfg.audio_codec.sample_rate
    wav = torch.stack([synthetic_clip(i, n) for i in range(4)]).to(dev)
    torch.set_default_device(dev)
    pad = torch.ones(B, T, dtype=torch.bool)
    video = torch.zeros(B, cfg.vision_encoder.dim, T)

    def field(t, y):
        return O.samaudio_forward(sd, cfg, y, feats, text, t.expand(B), video=video, text_mask=tmask, anchor_ids=ids,
                                  anchor_alignment=align, pad_mask=pad)

    out = {"clips": clips, "size": size}

    def one_step():
        return O.ode_fixed_grid(field, noise, method="midpoint", step_size=1.0)

    with torch.inference_mode():
        for label, ctx in (("fp32", torch.autocast("cuda", enabled=False)),
                           ("bf16_autocast", torch.autocast("cuda", dtype=torch.bfloat16))):
            with ctx:
                t0 = time.perf_counter()
                one_step()
                torch.cuda.synchronize()
                first = time.perf_counter() - t0
                sec = ev_time(one_step, iters=2, warm=1)
            out[label] = {"midpoint_step_s": sec, "first_call_s": first, "ode_16_steps_s": 16 * sec}
            print(f"eager {label}: one midpoint step (2 DiT evaluations, {clips} clips) {sec * 1e3:.1f} ms -> 16 steps "
                  f"{16 * sec:.2f} s (first call {first:.1f} s)", flush=True)
        # DAC-VAE through MIOpen: 4 waveforms, scaled to the batch (encode: 1 per clip, decode: 2 per clip)
        try:
            t0 = time.perf_counter()
            zz = O.dac_encode(sd, cfg.audio_codec, wav)
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            t_enc = ev_time(lambda: O.dac_encode(sd, cfg.audio_codec, wav), iters=2, warm=0)
            t_dec = ev_time(lambda: O.dac_decode(sd, cfg.audio_codec, zz), iters=2, warm=1)
            out["codec_fp32"] = {"encode_4_s": t_enc, "decode_4_s": t_dec, "first_call_s": first,
                                 "per_step_s": t_enc * clips / 4 + t_dec * 2 * clips / 4}
            print(f"eager DAC fp32: encode x4 {t_enc * 1e3:.1f} ms, decode x4 {t_dec * 1e3:.1f} ms -> "
                  f"{out['codec_fp32']['per_step_s']:.2f} s per {clips}-clip step (first call {first:.1f} s)", flush=True)
        except Exception as exc:  # MIOpen may lack a solver for a shape offline
            out["codec_fp32"] = {"error": repr(exc)}
            print("eager DAC failed:", exc, flush=True)
    codec_s = out.get("codec_fp32", {}).get("per_step_s")
    for label in ("fp32", "bf16_autocast"):
        tot = out[label]["ode_16_steps_s"] + (codec_s or 0.0)
        out[label]["separate_s"] = tot
        out[label]["s_audio_per_s"] = clips * 10.0 / tot
        print(f"eager {label}: separate() of {clips} clips ~ {tot:.2f} s = {out[label]['s_audio_per_s']:.1f} s-audio/s "
              f"(codec {'included, fp32' if codec_s else 'NOT included'})", flush=True)
    torch.set_default_device("cpu")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm", "eager", "all"])
    ap.add_argument("--size", default="large*")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "yardstick.json"))
    args = ap.parse_args()
    res = {}
    if os.path.exists(args.out):
        res = json.load(open(args.out))
    if args.what in ("gemm", "all"):
        res["gemm"] = gemm_yardstick(args.size)
    if args.what in ("eager", "all"):
        res["eager"] = eager_yardstick(args.size)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
