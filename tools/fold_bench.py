#!/usr/bin/env python
"""Timing of the folded cross-attention operand kernel (attention.hip cross_attn_fold_kernel) at the benchmark shape, in
the engine's own operand layout (V rows inside the all-layers K | V buffer: row stride L * 2D), HIP events on the stream.

    python tools/fold_bench.py [--clips 32]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from sam_audio_amd import hip, preset_config  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, nargs="+", default=[32, 16, 4])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    t = preset_config("large*").transformer
    D, H, L, Lt, ltp = t.dim, t.n_heads, t.n_layers, 8, 8
    KP = (H * ltp + 63) // 64 * 64
    L_ = hip.lib()
    g = torch.Generator(device=dev).manual_seed(1)
    wo = (torch.randn(D, D, generator=g, device=dev) / math.sqrt(D)).to(torch.bfloat16)
    for B in args.clips:
        kv_all = torch.randn(B * Lt, L * 2 * D, generator=g, device=dev).to(torch.bfloat16)
        kv = kv_all[:, 7 * 2 * D:]           # layer 7's K | V columns: a view with the full row stride
        ut = torch.zeros(B, D, KP, device=dev, dtype=torch.bfloat16)
        def run():
            hip.check(L_.samaudio_op_cross_attn_fold(hip.ptr(wo), kv.data_ptr(), L * 2 * D, hip.ptr(ut), KP, B, Lt, ltp, H,
                                                     hip.current_stream_ptr()))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 50
        v = kv_all[:, 7 * 2 * D + D: 8 * 2 * D].float().reshape(B, Lt, H, 128)
        ref = torch.einsum("nhd,bjhd->bnhj", wo.float().reshape(D, H, 128), v).reshape(B, D, H * Lt)
        err = (ut[:, :, :H * Lt].float() - ref).abs().max().item()
        out_mb = B * D * KP * 2 / 1e6
        print(f"cross_attn_fold B={B}: {us:7.1f} us  ({out_mb:.1f} MB out + {D * D * 2 / 1e6:.1f} MB Wo -> "
              f"{(out_mb + D * D * 2 / 1e6) / us * 1e-3:.2f} TB/s algorithmic)  max-abs err vs fp32 einsum {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
