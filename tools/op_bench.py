#!/usr/bin/env python
"""Times the non-GEMM kernels of one DiT layer at the benchmark shape (B=32, T=250, large* dims) with HIP events on
the launch stream; prints achieved GB/s against the bytes each kernel must move."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from sam_audio_amd.config import preset_config  # noqa: E402

dev = torch.device("cuda:0")
L = hip.lib()
t = preset_config("large*").transformer
B, T, H, D, Lt = 32, 250, t.n_heads, t.dim, 8
Tp, M = 256, B * T
st = hip.current_stream_ptr


def timeit(name, fn, nbytes, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"{name:32s} {us:8.1f} us   {nbytes / us / 1e3:7.1f} GB/s  ({nbytes / 1e6:.0f} MB algorithmic)", flush=True)


qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
qw, kw = torch.rand(128, device=dev) + 0.5, torch.rand(128, device=dev) + 0.5
ang = torch.rand(10000, 64, device=dev)
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
Q = torch.empty(B, H, Tp, 128, device=dev, dtype=torch.bfloat16)
K, Vt = torch.empty_like(Q), torch.empty(B, H, 128, Tp, device=dev, dtype=torch.bfloat16)
mask = torch.ones(B, T, dtype=torch.uint8, device=dev)
out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)


def prep():
    hip.check(L.samaudio_op_qkv_prep(hip.ptr(qkv), hip.ptr(qw), hip.ptr(kw), hip.ptr(cos), hip.ptr(sin), hip.ptr(Q),
                                     hip.ptr(K), hip.ptr(Vt), hip.BF16, B, T, Tp, H, 1e-5, st()))


for flag, name in ((1, "qkv_prep (first generation)"), (0, "qkv_prep (16-byte accesses)")):
    L.samaudio_debug_set_flag(1, flag)
    timeit(name, prep, 2 * M * 3 * D * 2)
L.samaudio_debug_set_flag(1, 0)
for flag, name in ((0, "self_attention (8 waves, 128 query rows)"), (1, "self_attention (16 waves, 256 query rows)")):
    L.samaudio_debug_set_flag(13, flag)
    timeit(name, lambda: hip.check(L.samaudio_op_self_attention(
        hip.ptr(Q), hip.ptr(K), hip.ptr(Vt), hip.ptr(mask), hip.ptr(out), hip.BF16, B, T, Tp, H, st())), 4 * M * D * 2)
L.samaudio_debug_set_flag(13, 0)
L.samaudio_debug_set_flag(23, 1)
timeit("self_attention (8 waves, query blocks of a (batch, head) on one XCD)", lambda: hip.check(L.samaudio_op_self_attention(
    hip.ptr(Q), hip.ptr(K), hip.ptr(Vt), hip.ptr(mask), hip.ptr(out), hip.BF16, B, T, Tp, H, st())), 4 * M * D * 2)
L.samaudio_debug_set_flag(23, 0)
x = torch.randn(M, D, device=dev)
w = torch.rand(D, device=dev)
tab = torch.randn(6, D, device=dev)
t0 = torch.randn(1, 6 * D, device=dev)
xn = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
for flag, name in ((1, "rmsnorm_mod (two-pass, round 1)"), (0, "rmsnorm_mod (row in registers)")):
    L.samaudio_debug_set_flag(2, flag)
    timeit(name, lambda: hip.check(L.samaudio_op_rmsnorm_mod(
        hip.ptr(x), hip.ptr(w), C.c_void_p(tab[0].data_ptr()), C.c_void_p(tab[1].data_ptr()), hip.ptr(t0), 0, 0, D,
        hip.ptr(xn), hip.BF16, M, D, T, 1e-5, st())), M * D * 6)
L.samaudio_debug_set_flag(2, 0)
# probe: the same kernel without the modulation operands (4 of its 5 table vectors are not loaded): what do they cost?
timeit("rmsnorm without modulation (probe)", lambda: hip.check(L.samaudio_op_rmsnorm_mod(
    hip.ptr(x), hip.ptr(w), None, None, None, 0, 0, 0, hip.ptr(xn), hip.BF16, M, D, T, 1e-5, st())), M * D * 6)
q = torch.randn(M, D, device=dev).to(torch.bfloat16)
kv = torch.randn(B * Lt, 2 * D, device=dev).to(torch.bfloat16)
tmask = torch.ones(B, Lt, dtype=torch.uint8, device=dev)
timeit("cross_attention (+k headnorm)", lambda: hip.check(L.samaudio_op_cross_attention(
    hip.ptr(q), hip.ptr(qw), hip.ptr(kv), hip.ptr(kw), hip.ptr(tmask), hip.ptr(out), hip.BF16, B, T, Lt, H, 1e-5, st())),
    2 * M * D * 2)

wo = (torch.randn(D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
KP = (H * 8 + 63) // 64 * 64
ut = torch.zeros(B, D, KP, device=dev, dtype=torch.bfloat16)
for flag, name in ((0, "cross_attn_fold"), (1, "cross_attn_fold (LDS-staged rows, candidate)")):
    L.samaudio_debug_set_flag(3, flag)
    timeit(name, lambda: hip.check(L.samaudio_op_cross_attn_fold(hip.ptr(wo), hip.ptr(kv), 2 * D, hip.ptr(ut), KP, B, Lt,
                                                                 8, H, st())), B * D * KP * 2 + D * D * 2)
L.samaudio_debug_set_flag(3, 0)
for zs in (1, 2, 4):
    L.samaudio_debug_set_flag(12, zs)
    timeit(f"cross_attn_fold, batch split {zs}", lambda: hip.check(L.samaudio_op_cross_attn_fold(
        hip.ptr(wo), hip.ptr(kv), 2 * D, hip.ptr(ut), KP, B, Lt, 8, H, st())), B * D * KP * 2 + D * D * 2)
L.samaudio_debug_set_flag(12, 0)
# ... with the engine's real key/value layout: one [B*Lt, L*2D] tensor for all 22 layers (row stride 248 KB)
kv_all = torch.randn(B * Lt, 22 * 2 * D, device=dev).to(torch.bfloat16)
timeit("cross_attn_fold (kv_all layout)", lambda: hip.check(L.samaudio_op_cross_attn_fold(
    hip.ptr(wo), C.c_void_p(kv_all.data_ptr() + 5 * 2 * D * 2), 22 * 2 * D, hip.ptr(ut), KP, B, Lt, 8, H, st())),
    B * D * KP * 2 + D * D * 2)

# DAC decoder stage with 192 channels: dilated k7 conv as implicit GEMM (N = 192, K = 1344), 8 waveforms
from tests import util  # noqa: E402
items, Tc, Cc = 8, 240000, 192
xa = torch.randn(items, Tc + 80, Cc, device=dev).to(torch.bfloat16)
wc = (torch.randn(Cc, 7 * Cc, device=dev) / (7 * Cc) ** 0.5).to(torch.bfloat16)
oc = torch.empty(items, Tc + 80, Cc, device=dev, dtype=torch.bfloat16)
bias_c, alpha_c = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
for flag, name in ((1, "codec conv7 C=192 (2 x 128-wide tiles)"), (0, "codec conv7 C=192 (256x192 tile)"), (34, "codec conv7 C=192 (128x192 k32 s3, 2 wg/CU)"), (35, "codec conv7 C=192 (conv7h: halo tile resident)")):
    L.samaudio_debug_set_flag(4, 1 if flag == 1 else 0)
    L.samaudio_debug_force_gemm_variant(flag if flag > 1 else -1)
    timeit(name, lambda: util.gemm("bf16", xa, wc, Tc, Cc, 7 * Cc, nbatch=items, a_off=(40 - 9) * Cc, a_bstride=(Tc + 80) * Cc,
                                   lda=Cc, kc=Cc, tap_stride=3 * Cc, bias=bias_c, out_act=oc,
                                   act_geom=((Tc + 80) * Cc, Cc, 40 * Cc), act=hip.ACT_SNAKE, act_alpha=alpha_c),
           2 * items * Tc * Cc * 2, iters=5)
L.samaudio_debug_set_flag(4, 0)
L.samaudio_debug_force_gemm_variant(-1)

# the k1 convolution that closes a residual unit: raw (fp32) += W x, bf16 copy snake(raw) for the next unit - HBM-bound
raw = torch.randn(items, Tc + 80, Cc, device=dev)
w1 = (torch.randn(Cc, Cc, device=dev) / Cc ** 0.5).to(torch.bfloat16)
for v192 in (-1, 34, 29):
  L.samaudio_debug_force_gemm_variant(v192)
  timeit(f"codec conv1 C=192 + in-place residual [variant {v192}]", lambda: util.gemm(
    "bf16", xa, w1, Tc, Cc, Cc, nbatch=items, a_off=40 * Cc, a_bstride=(Tc + 80) * Cc, lda=Cc, bias=bias_c, res=raw,
    res_geom=((Tc + 80) * Cc, Cc, 40 * Cc), out_f32=raw, f32_geom=((Tc + 80) * Cc, Cc, 40 * Cc), out_act=oc,
    act_geom=((Tc + 80) * Cc, Cc, 40 * Cc), act=hip.ACT_SNAKE, act_alpha=alpha_c), items * Tc * Cc * (2 + 4 + 4 + 2), iters=5)
L.samaudio_debug_force_gemm_variant(-1)

# the last decoder stage (96 channels, T = 480 000): which tile family suits a 96-wide output?  (8 waveforms)
items, Tc, Cc = 8, 480000, 96
xa = torch.randn(items, Tc + 80, Cc, device=dev).to(torch.bfloat16)
Kp = (7 * Cc + 63) // 64 * 64
wc = (torch.randn(Cc, Kp, device=dev) / (7 * Cc) ** 0.5).to(torch.bfloat16)
oc = torch.empty(items, Tc + 80, Cc, device=dev, dtype=torch.bfloat16)
raw = torch.randn(items, Tc + 80, Cc, device=dev)
w1 = (torch.randn(Cc, 128, device=dev) / Cc ** 0.5).to(torch.bfloat16)   # K padded to 128
bias_c, alpha_c = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
for v, vname in ((-1, "policy"), (29, "128x128 k32 s3 (3 wg/CU)"), (35, "conv7h: halo tile resident"), (33, "64x128 k32 s3 (4 wg/CU)")):
    L.samaudio_debug_force_gemm_variant(v)
    timeit(f"codec conv7 C=96 [{vname}]", lambda: util.gemm(
        "bf16", xa, wc, Tc, Cc, Kp, nbatch=items, a_off=(40 - 3) * Cc, a_bstride=(Tc + 80) * Cc, lda=Cc, kc=Cc, tap_stride=Cc,
        bias=bias_c, out_act=oc, act_geom=((Tc + 80) * Cc, Cc, 40 * Cc), act=hip.ACT_SNAKE, act_alpha=alpha_c),
        2 * items * Tc * Cc * 2, iters=5)
    timeit(f"codec conv1 C=96 + residual [{vname}]", lambda: util.gemm(
        "bf16", xa, w1, Tc, Cc, 128, nbatch=items, a_off=40 * Cc, a_bstride=(Tc + 80) * Cc, lda=Cc, kc=Cc, bias=bias_c, res=raw,
        res_geom=((Tc + 80) * Cc, Cc, 40 * Cc), out_f32=raw, f32_geom=((Tc + 80) * Cc, Cc, 40 * Cc), out_act=oc,
        act_geom=((Tc + 80) * Cc, Cc, 40 * Cc), act=hip.ACT_SNAKE, act_alpha=alpha_c), items * Tc * Cc * (2 + 4 + 4 + 2), iters=5)
L.samaudio_debug_force_gemm_variant(-1)

# conv7h vs the implicit GEMM the policy would pick without it (flag 11), at the other channel counts conv7h covers
for Cc, Tc in ((64, 480000), (128, 240000)):
    items = 8
    xa = torch.randn(items, Tc + 80, Cc, device=dev).to(torch.bfloat16)
    wc = (torch.randn(Cc, 7 * Cc, device=dev) / (7 * Cc) ** 0.5).to(torch.bfloat16)
    oc = torch.empty(items, Tc + 80, Cc, device=dev, dtype=torch.bfloat16)
    bias_c, alpha_c = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    for f11, vname in ((0, "policy: conv7h"), (1, "policy without conv7h")):
        L.samaudio_debug_set_flag(11, f11)
        timeit(f"codec conv7 C={Cc} dil 3 [{vname}]", lambda: util.gemm(
            "bf16", xa, wc, Tc, Cc, 7 * Cc, nbatch=items, a_off=(40 - 9) * Cc, a_bstride=(Tc + 80) * Cc, lda=Cc, kc=Cc,
            tap_stride=3 * Cc, bias=bias_c, out_act=oc, act_geom=((Tc + 80) * Cc, Cc, 40 * Cc), act=hip.ACT_SNAKE,
            act_alpha=alpha_c), 2 * items * Tc * Cc * 2, iters=5)
    L.samaudio_debug_set_flag(11, 0)
items, Tc, Cc = 8, 480000, 96

# one DAC residual unit: the two launches (k7 + k1, tile policy) against the fused resunit kernel
import ctypes as CT  # noqa: E402
for Cc, Tc, items in ((64, 480000, 8), (96, 480000, 8), (128, 240000, 8), (192, 240000, 8)):
    geom = ((Tc + 80) * Cc, Cc, 40 * Cc)
    K7, K1 = (7 * Cc + 63) // 64 * 64, (Cc + 63) // 64 * 64
    xa = torch.randn(items, Tc + 80, Cc, device=dev).to(torch.bfloat16)
    mid = torch.zeros_like(xa)
    oc = torch.zeros_like(xa)
    raw = torch.randn(items, Tc + 80, Cc, device=dev)
    w7 = (torch.randn(Cc, K7, device=dev) / (7 * Cc) ** 0.5).to(torch.bfloat16)
    w1 = (torch.randn(Cc, K1, device=dev) / Cc ** 0.5).to(torch.bfloat16)
    bias_c, alpha_c = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    p7 = util.gemm_params(xa, w7, Tc, Cc, K7, nbatch=items, a_off=(40 - 9) * Cc, a_bstride=geom[0], lda=Cc, kc=Cc,
                          tap_stride=3 * Cc, bias=bias_c, out_act=mid, act_geom=geom, act=hip.ACT_SNAKE, act_alpha=alpha_c)
    p1 = util.gemm_params(mid, w1, Tc, Cc, K1, nbatch=items, a_off=40 * Cc, a_bstride=geom[0], lda=Cc, kc=Cc, tap_stride=Cc,
                          bias=bias_c, res=raw, res_geom=geom, out_f32=raw, f32_geom=geom, out_act=oc, act_geom=geom,
                          act=hip.ACT_SNAKE, act_alpha=alpha_c)
    unit_bytes = items * Tc * Cc * (2 + 4 + 4 + 2)

    def two():
        for q in (p7, p1):
            hip.check(L.samaudio_op_gemm(CT.byref(q), CT.sizeof(q), hip.BF16, st()))
    timeit(f"residual unit C={Cc} dil 3 [two launches]", two, unit_bytes, iters=5)
    timeit(f"residual unit C={Cc} dil 3 [fused resunit]", lambda: util.resunit(p7, p1), unit_bytes, iters=5)
    if Cc == 96:
        L.samaudio_debug_set_flag(20, 1)
        timeit(f"residual unit C={Cc} dil 3 [fused, 256-row tiles on 8 waves, 1 wg/CU]", lambda: util.resunit(p7, p1), unit_bytes, iters=5)
        L.samaudio_debug_set_flag(20, 0)
items, Tc, Cc = 8, 480000, 96

# the same contraction as a PLAIN GEMM (dense A [M, 704]): separates the implicit-convolution addressing from the
# narrow-N / short-K regime
Mp = items * Tc
Ad = torch.randn(Mp // 4, Kp, device=dev).to(torch.bfloat16)      # a quarter of the rows (5.4 GB would not fit the timing loop comfortably)
od = torch.empty(Mp // 4, Cc, device=dev, dtype=torch.bfloat16)
for v, vname in ((20, "ld 256x128 persist"), (25, "128x128 s2"), (4, "ring 256x128 s2")):
    L.samaudio_debug_force_gemm_variant(v)
    timeit(f"plain GEMM M={Mp // 4} N=96 K={Kp} [{vname}] (x4 = conv7 C=96)", lambda: util.gemm(
        "bf16", Ad, wc, Mp // 4, Cc, Kp, bias=bias_c, out_act=od, act_geom=(0, Cc, 0), act=hip.ACT_SNAKE, act_alpha=alpha_c),
        (Mp // 4) * (Kp + Cc) * 2, iters=5)
L.samaudio_debug_force_gemm_variant(-1)
