#!/usr/bin/env python
"""What the 8-phase K loop spends its time on: the ablation builds of the round-3 loop gemm8o_kernel (tools/build_abl.sh; debug flag 25: 8 = that loop unchanged, 1 = no
DMA in the loop, 2 = no LDS fragment reads, 3 = no MFMA, 4 = no barriers, 5 = no s_setprio) timed against K at fixed (M, N)
- slope = time per K-tile of what is left - and the s_memtime stamps of the real kernel (flag 25 = 9): cycles from kernel
entry to the end of the prologue, through the K loop, through the epilogue, per tile.

    SAMAUDIO_LIB_AB=tools/abl/libsamaudio_hip_abl.so python tools/gemm8_ablate.py
"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from tests import util  # noqa: E402

NAMES = {0: "shipped kernel", 8: "round-3 loop", 1: "r3 loop, no DMA", 2: "r3 loop, no LDS reads", 3: "r3 loop, no MFMA",
         4: "r3 loop, no barriers", 5: "r3 loop, no setprio"}


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    assert os.environ.get("SAMAUDIO_LIB_AB"), "run with SAMAUDIO_LIB_AB=tools/abl/libsamaudio_hip_abl.so"
    dev = torch.device("cuda:0")
    lib = hip.lib()
    lib.samaudio_debug_force_gemm_variant(22)
    Ks = [1024, 2816, 5632, 8448]
    for (M, N) in [(4000, 2816), (4096, 4096)]:
        print(f"M={M} N={N} plain 16-bit output; us per launch at K = {Ks}; slope over K")
        for abl in (0, 8, 1, 2, 3, 4, 5):
            us = []
            for K in Ks:
                A = torch.randn(M, K, device=dev).to(torch.bfloat16)
                W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
                out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                lib.samaudio_debug_set_flag(25, abl)
                us.append(timeit(lambda: util.gemm("bf16", A, W, M, N, K, out_act=out, act_geom=(0, N, 0))))
            lib.samaudio_debug_set_flag(25, 0)
            slope = (us[-1] - us[0]) / ((Ks[-1] - Ks[0]) / 64)
            print(f"  {NAMES[abl]:22s}: " + " ".join(f"{u:7.1f}" for u in us) + f" | {slope:.3f} us per K-tile", flush=True)
    # timestamps of the real kernel
    for (M, N, K, kind) in [(4000, 2816, 2816, "plain"), (4000, 8448, 2816, "plain"), (4000, 2816, 2816, "gated"), (4096, 4096, 4096, "plain")]:
        A = torch.randn(M, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        ts = torch.zeros(tiles * 8 * 4, dtype=torch.int64, device=dev)
        kw = dict(out_act=out, act_geom=(0, N, 0), act_alpha=ts)
        if kind == "gated":
            T = 250
            kw.update(gate_tab=torch.randn(N, device=dev), gate=torch.randn((M + T - 1) // T, N, device=dev), gate_ld=N,
                      rows_per_gate=T, res=torch.randn(M, N, device=dev), res_geom=(0, N, 0), out_f32=torch.empty(M, N, device=dev),
                      f32_geom=(0, N, 0))
        lib.samaudio_debug_set_flag(25, 9)
        for _ in range(3):
            util.gemm("bf16", A, W, M, N, K, **kw)
        torch.cuda.synchronize()
        lib.samaudio_debug_set_flag(25, 0)
        t = ts.view(tiles, 8, 4).cpu().double()
        t0 = t[..., 0].min()            # first wave to enter, any tile (comparable across CUs only if the counter is chip-wide)
        pro, loop, epi = (t[..., 1] - t[..., 0]), (t[..., 2] - t[..., 1]), (t[..., 3] - t[..., 2])
        start = t[..., 0] - t0
        end = t[..., 3] - t0
        tick = 1e-3   # s_memtime counts shader cycles: report kilo-cycles
        print(f"M={M} N={N} K={K} {kind}: {tiles} tiles; per wave, kilo-cycles of s_memtime (mean / max): entry offset {start.mean() * tick:.2f} / {start.max() * tick:.2f}, "
              f"prologue {pro.mean() * tick:.2f} / {pro.max() * tick:.2f}, K loop {loop.mean() * tick:.2f} / {loop.max() * tick:.2f}, "
              f"epilogue {epi.mean() * tick:.2f} / {epi.max() * tick:.2f}, last wave done at {end.max() * tick:.2f}", flush=True)
    lib.samaudio_debug_force_gemm_variant(-1)


if __name__ == "__main__":
    main()
