#!/usr/bin/env python
"""Times the K' = 3K launches of the fp16x3 mode on the real DiT shapes, plain walk against the operand-sharing walk
(GemmParams.flags bit 15, gemm8.hip gemm8x_kernel / gemm8s_kernel's index map).  usage: [PROBE_ROWS=4000] python tools/x3_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from sam_audio_amd.config import preset_config  # noqa: E402
from tests import util  # noqa: E402

t = preset_config("large*").transformer
D, Fh = t.dim, t.ffn_hidden
dev = torch.device("cuda:0")
SUSTAIN = float(os.environ.get("PROBE_SUSTAIN", "0"))   # seconds of back-to-back w13 launches (shared walk) per shape: for clock / power sampling beside it
for M in [int(r) for r in os.environ.get("PROBE_ROWS", "4000,500").split(",")]:
    shapes = {"qkv": (M, 3 * D, D), "wo": (M, D, D), "w13": (M, 2 * Fh, D), "w2": (M, D, Fh)}
    for name, (m, n, k) in shapes.items():
        k3 = 3 * k
        x, w = torch.randn(m, k, device=dev), torch.randn(n, k, device=dev) / k ** 0.5   # the split forms of sam_audio_amd.weights.x3_weight
        xh, wh = x.half(), w.half()
        A = torch.cat([(x - xh.float()).half(), xh, xh], dim=1).contiguous()
        W = torch.cat([wh, (w - wh.float()).half(), wh], dim=1).contiguous()
        del x, w, xh, wh
        out = torch.empty(m, n, device=dev, dtype=torch.float32)
        line = f"M={m} {name} N={n} K'={k3}:"
        keep = {}
        for flags in (0, 32768):
            if k % 64:
                continue
            for _ in range(2):
                util.gemm("fp16", A, W, m, n, k3, out_f32=out, f32_geom=(0, n, 0), flags=flags)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(10):
                util.gemm("fp16", A, W, m, n, k3, out_f32=out, f32_geom=(0, n, 0), flags=flags)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            keep[flags] = out.clone()
            line += f"  {'shared' if flags else 'plain'} {us:8.1f} us {2.0 * m * n * k3 / us / 1e6:7.1f} TF/s(mfma)"
        if len(keep) == 2:   # another order of the same sum: fp32 rounding apart (a race or a missed wait shows as O(1))
            line += f"  |shared - plain| <= {(keep[0] - keep[32768]).abs().max().item():.2e} (|out| <= {keep[0].abs().max().item():.1f})"
        if 32768 in keep:   # the full-size race check: ten more launches and the 128 x 128 kernel's walk of the same order, bit for bit
            lib = hip.lib(hip.operands_for("fp16"))
            same = True
            for _ in range(10):
                util.gemm("fp16", A, W, m, n, k3, out_f32=out, f32_geom=(0, n, 0), flags=32768)
                same &= torch.equal(out, keep[32768])
            lib.samaudio_debug_force_gemm_variant(27)
            try:
                out.fill_(float("nan"))
                util.gemm("fp16", A, W, m, n, k3, out_f32=out, f32_geom=(0, n, 0), flags=32768)
            finally:
                lib.samaudio_debug_force_gemm_variant(-1)
            line += f"  repeatable {same}, gemm8s bitwise {torch.equal(out, keep[32768])}"
        print(line, flush=True)
        if SUSTAIN > 0 and name in os.environ.get("PROBE_SUSTAIN_SHAPES", "w13").split(","):
            import time
            t0 = time.time()
            n_l = 0
            while time.time() - t0 < SUSTAIN:
                for _ in range(200):
                    util.gemm("fp16", A, W, m, n, k3, out_f32=out, f32_geom=(0, n, 0), flags=32768)
                torch.cuda.synchronize()
                n_l += 200
            dt = time.time() - t0
            print(f"  sustained {name}: {n_l} launches in {dt:.1f} s = {dt / n_l * 1e6:.1f} us per launch, {2.0 * m * n * k3 * n_l / dt / 1e12:.1f} TF/s(mfma)", flush=True)
