#!/usr/bin/env python
"""Time of one 8-phase launch against K at fixed (M, N): separates the per-K-tile time of the K loop (slope) from the
per-tile fixed cost - launch, prologue latency, epilogue (intercept).  Forced variant 22 (256x256), one round of tiles for
the first two shapes.  Each debug-flag setting given with --flags is one column (e.g. --flags 23=0 23=1).

    python tools/gemm_ksweep.py [--flags 23=0 23=1] [--iters 20]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from tests import util  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", nargs="*", default=["23=0", "23=1"])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--variant", type=int, default=22)
    ap.add_argument("--kinds", nargs="*", default=["plain", "gated"])
    ap.add_argument("--shapes", nargs="*", default=["4000x2816", "4096x4096", "4000x8448", "8000x2816"])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(3)
    Ks = [64, 128, 256, 512, 1024, 2048, 2816, 4096, 5632, 8448]
    for (M, N) in [tuple(int(v) for v in sh.split("x")) for sh in args.shapes]:
        for kind in args.kinds:
            rows = {}
            for K in Ks:
                A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
                W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
                out_act = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                kw = dict(out_act=out_act, act_geom=(0, N, 0))
                if kind == "gated":
                    T = 250
                    B = (M + T - 1) // T
                    kw.update(gate_tab=torch.randn(N, device=dev), gate=torch.randn(B, N, device=dev), gate_ld=N,
                              rows_per_gate=T, res=torch.randn(M, N, device=dev), res_geom=(0, N, 0),
                              out_f32=torch.empty(M, N, device=dev), f32_geom=(0, N, 0))
                for fl in args.flags:
                    k_, v_ = fl.split("=")
                    hip.lib().samaudio_debug_force_gemm_variant(args.variant)
                    hip.lib().samaudio_debug_set_flag(int(k_), int(v_))
                    for _ in range(3):
                        util.gemm("bf16", A, W, M, N, K, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.iters):
                        util.gemm("bf16", A, W, M, N, K, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    rows.setdefault(fl, []).append(e0.elapsed_time(e1) * 1e3 / args.iters)
                    hip.lib().samaudio_debug_set_flag(int(k_), 0)
            hip.lib().samaudio_debug_force_gemm_variant(-1)
            tiles = ((M + 255) // 256) * ((N + 255) // 256)
            print(f"M={M} N={N} {kind}: {tiles} tiles; K = {Ks}")
            for fl, us in rows.items():
                # least-squares slope / intercept over K >= 1024 (steady state), in us per K-tile of 64 and us
                xs = [K / 64 for K in Ks if K >= 1024]
                ys = [u for K, u in zip(Ks, us) if K >= 1024]
                n = len(xs)
                sx, sy = sum(xs), sum(ys)
                sxx, sxy = sum(x * x for x in xs), sum(x * y for x, y in zip(xs, ys))
                slope = (n * sxy - sx * sy) / (n * sxx - sx * sx)
                icpt = (sy - slope * sx) / n
                print(f"  flags {fl}: " + " ".join(f"{u:7.1f}" for u in us) + f" us | {slope:.3f} us/K-tile + {icpt:.1f} us", flush=True)


if __name__ == "__main__":
    main()
