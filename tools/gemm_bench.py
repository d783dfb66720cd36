#!/usr/bin/env python
"""GEMM sweep on the real DiT shapes (run on the GPU box): correctness of the policy's kernels against a torch fp32 matmul
on the same bf16-rounded operands, then timing (HIP events on the launch stream, random data), with hipBLASLt
(`torch.matmul`, plain product) on the same operands beside it.

    python tools/gemm_bench.py [--dims large*|default|small*] [--clips 32 16 4] [--iters 10]
"""
import argparse
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from sam_audio_amd.config import preset_config  # noqa: E402
from tests import util  # noqa: E402

RASTER = [0]
BLAS = [False]
# the kernels the tile policy picks from for a DiT-class GEMM (gemm.hip gemm_variant numbering)
VARIANTS = {-1: "auto policy (tail split)", 22: "8-phase 256x256", 27: "gemm8s 128x128"}
# A/B of a debug flag on the forced 8-phase kernel: --ab FLAG adds a column "8-phase, flag FLAG = 1"
AB_FLAG = [None]
ROLES = [False]
AB_VALUE = [1]


def interleave16(w1, w3):
    F_, K = w1.shape
    return torch.stack([w1.view(F_ // 16, 16, K), w3.view(F_ // 16, 16, K)], 1).reshape(2 * F_, K)


def run_case(name, M, N, K, kind, dev, iters, T=250):
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    flops = 2.0 * M * N * K
    kw = {}
    if kind == "swiglu":
        w1 = (torch.randn(N // 2, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        w3 = (torch.randn(N // 2, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        W = interleave16(w1, w3).contiguous()
        ref = (F.silu(A.float() @ w1.float().T) * (A.float() @ w3.float().T))
        out_act = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
        kw = dict(swiglu=1, out_act=out_act, act_geom=(0, N // 2, 0))
        outs = [("act", out_act, ref, 6e-2)]
    else:
        W = (torch.randn(N, K, generator=g, device=dev) / math.sqrt(K)).to(torch.bfloat16)
        base = A.float() @ W.float().T
        if kind == "plain":
            out_act = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            kw = dict(out_act=out_act, act_geom=(0, N, 0))
            outs = [("act", out_act, base, 6e-2)]
        else:  # gated residual, fp32 + bf16 outputs (wo / w2 epilogue)
            B = (M + T - 1) // T
            tab = torch.randn(N, generator=g, device=dev)
            gate = torch.randn(B, 6 * N, generator=g, device=dev)
            res = torch.randn(M, N, generator=g, device=dev)
            out = torch.empty(M, N, device=dev)
            out_act = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            rows = torch.arange(M, device=dev) // T
            ref = base * (tab[None] + gate[rows, 2 * N:3 * N]) + res
            gsl = gate[:, 2 * N:]
            kw = dict(gate_tab=tab, gate=gsl, gate_ld=6 * N, rows_per_gate=T, res=res, res_geom=(0, N, 0), out_f32=out,
                      f32_geom=(0, N, 0), out_act=out_act, act_geom=(0, N, 0))
            outs = [("f32", out, ref, 2e-3), ("act", out_act, ref, 1e-1)]
    kw["raster_gm"] = RASTER[0]
    line = f"{name:>22s} M={M} N={N} K={K} {kind:7s} gm={RASTER[0]}"
    cols = [(v, vname, None) for v, vname in VARIANTS.items()]
    if AB_FLAG[0] is not None:
        cols += [(22, f"8-phase flag {AB_FLAG[0]}={AB_VALUE[0]}", (AB_FLAG[0], AB_VALUE[0])),
                 (-1, f"auto flag {AB_FLAG[0]}={AB_VALUE[0]}", (AB_FLAG[0], AB_VALUE[0]))]
    if ROLES[0]:   # the pipelined gemm8s form's wave roles (debug flag 27: 1 = none, 2 / 3 = requesting waves, PROD 0 / 2)
        cols = [(27, f"gemm8s flag 27={r}", (27, r)) for r in (1, 2, 3)] + [(22, "8-phase 256x256", None)]
    for v, vname, flag in cols:
        hip.lib().samaudio_debug_force_gemm_variant(v)
        if flag is not None:
            hip.lib().samaudio_debug_set_flag(flag[0], flag[1])
        for _, o, _, _ in outs:
            o.fill_(float("nan"))
        util.gemm("bf16", A, W, M, N, K, **kw)
        torch.cuda.synchronize()
        errs = []
        ok = True
        for oname, o, ref, tol in outs:
            e = (o.float() - ref).abs().max().item()
            errs.append(f"{oname} {e:.1e}")
            ok &= e <= tol and bool(torch.isfinite(o.float()).all())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            util.gemm("bf16", A, W, M, N, K, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        line += f" | {vname}: {us:8.1f} us {flops / us / 1e6:7.1f} TF {'ok' if ok else 'WRONG'} ({', '.join(errs)})"
        if flag is not None:
            hip.lib().samaudio_debug_set_flag(flag[0], 0)
    hip.lib().samaudio_debug_force_gemm_variant(-1)
    if BLAS[0]:   # yardstick: hipBLASLt through torch.matmul on the same operands (plain product, 16-bit output, no epilogue)
        for _ in range(2):
            torch.matmul(A, W.t())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            torch.matmul(A, W.t())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        line += f" | hipBLASLt (plain): {us:8.1f} us {flops / us / 1e6:7.1f} TF"
    print(line, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dims", default="large*")
    ap.add_argument("--clips", type=int, nargs="+", default=[32, 16, 4], help="rows = clips x 250 frames")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--quick", action="store_true", help="only the small correctness shapes")
    ap.add_argument("--no-blas", action="store_true", help="skip the hipBLASLt (torch.matmul) column")
    ap.add_argument("--raster", action="store_true", help="sweep the tile-raster group size of the 8-phase kernel")
    ap.add_argument("--ab", type=int, default=None, help="debug flag to A/B on the forced 8-phase kernel and the policy")
    ap.add_argument("--ab-value", type=int, default=1)
    ap.add_argument("--roles", action="store_true", help="columns: gemm8s under debug flag 27 = 1 / 2 / 3 (wave roles of the pipelined form)")
    ap.add_argument("--vit", action="store_true", help="the PE-Core-L14-336 tower's GEMM shapes (250 frames x 577 tokens)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    BLAS[0] = not args.no_blas
    AB_FLAG[0] = args.ab
    AB_VALUE[0] = args.ab_value
    ROLES[0] = args.roles
    if args.vit:
        Mv = 250 * 577
        run_case("vit qkv (bias)", Mv, 3072, 1024, "plain", dev, 5)
        run_case("vit out_proj (bias+res)", Mv, 1024, 1024, "gated", dev, 5)
        run_case("vit c_fc (bias+gelu)", Mv, 4096, 1024, "plain", dev, 5)
        run_case("vit c_proj (bias+res)", Mv, 1024, 4096, "gated", dev, 5)
        return
    t = preset_config(args.dims).transformer
    D, Fh = t.dim, t.ffn_hidden
    # small correctness shapes first: ragged M, N tails, several K
    run_case("edge small", 300, 640, 192, "plain", dev, 2)
    run_case("edge gated", 517, 1152, 320, "gated", dev, 2, T=47)
    run_case("edge swiglu", 333, 1280, 256, "swiglu", dev, 2)
    if args.quick:
        return
    # the guide's reference shape (cdna_hip_programming.md section 5: 8-phase template ~1320-1340 TF at 4096^3 on random
    # operands): calibrates this box / this kernel against that ladder
    run_case("square 4096", 4096, 4096, 4096, "plain", dev, args.iters)
    for gm in ((8, 2, 4, 16, 32) if args.raster else (0,)):
        RASTER[0] = gm
        for clips in args.clips:
            M = clips * 250
            run_case("qkv", M, 3 * D, D, "plain", dev, args.iters)
            run_case("wo/c_wo (gate+res)", M, D, D, "gated", dev, args.iters)
            run_case("c_wq", M, D, D, "plain", dev, args.iters)
            run_case("w13 swiglu", M, 2 * Fh, D, "swiglu", dev, args.iters)
            run_case("w2 (gate+res)", M, D, Fh, "gated", dev, args.iters)
            run_case("w2 shape, plain (as hipBLASLt)", M, D, Fh, "plain", dev, args.iters)
            run_case("patcher conv-as-gemm shape", M, D, 3 * D, "plain", dev, args.iters)


if __name__ == "__main__":
    main()
