#!/usr/bin/env python
"""Times the fp32 kernel's compensated launches (GemmParams.flags bits 13 / 14, gemm.hip) on the shapes of the DAC-VAE stages with
< 256 channels, one launch per shape and form, with torch events on the launch stream.
usage: [PROBE_ROWS=1920000] python tools/fly_probe.py        forms: register split of both operands on the plain tiles | split weight
on the plain tiles | split weight on the 4 x 1-wave tiles"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from sam_audio_amd.weights import fly16_weight  # noqa: E402
from tests import util  # noqa: E402

M = int(os.environ.get("PROBE_ROWS", "1920000"))   # 4 waveforms of 10 s at 48 kHz
dev = torch.device("cuda:0")
lib = hip.lib("fp16")
# (name, N, K, kc, dilation)
SHAPES = [("k7 96ch d1", 96, 672, 96, 1), ("k7 96ch d9", 96, 672, 96, 9), ("k1 96ch", 96, 96, 96, 1), ("k7 192ch d3", 192, 1344, 192, 3),
          ("k1 192ch", 192, 192, 192, 1), ("up 192->2x96", 192, 384, 384, 1), ("k7 128ch d1", 128, 896, 128, 1), ("k7 64ch d1", 64, 448, 64, 1),
          ("k7 96->1", 1, 672, 96, 1)]
FORMS = [("fly", 8192, 0), ("twin old tiles", 8192 | 16384, 1), ("twin 4x1", 8192 | 16384, 0)]
for name, N, K, kc, dil in SHAPES:
    rows = M // 2 if N == 192 and kc == 192 else M    # the 192-channel stage runs at half the sample rate
    taps, halo = K // kc, 32
    x = torch.randn(rows + 2 * halo, kc, device=dev)
    w = torch.randn(N, K) * 0.05
    wd, wf = w.to(dev), fly16_weight(w, torch.float16).to(dev)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(rows, max(N, 4), device=dev)
    a_off = (halo - (taps // 2) * dil) * kc
    line = []
    for form, flags, dbg in FORMS:
        prm = util.gemm_params(x, wf if flags & 16384 else wd, rows, N, K, a_off=a_off, lda=kc, kc=kc, tap_stride=dil * kc if taps > 1 else 0,
                               bias=bias, out_f32=out, f32_geom=(0, out.shape[1], 0), flags=flags)
        lib.samaudio_debug_set_flag(36, dbg)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        for it in range(3):
            if it == 1:
                ev[0].record()
            hip.check(lib.samaudio_op_gemm(C.byref(prm), C.sizeof(prm), hip.F32, util.stream()))
        ev[1].record()
        torch.cuda.synchronize()
        lib.samaudio_debug_set_flag(36, 0)
        ms = ev[0].elapsed_time(ev[1]) / 2
        line.append(f"{form} {ms:7.3f} ms {2.0 * rows * N * K / ms / 1e9:6.1f} TF/s")
    print(f"{name:14s} rows {rows}: " + " | ".join(line) + f" | fp32 in+out {rows * (kc + N) * 4 / 1e9:.2f} GB", flush=True)
