#!/usr/bin/env python
"""Times the non-GEMM kernels of one DiT layer and the DAC-VAE residual units at the benchmark shape (B=32, T=250, large*
dims; 8 waveforms per codec launch) with HIP events on the launch stream; prints achieved GB/s against the bytes each
kernel must move.  (The A/B loops over retired kernel generations that this file carried in round 2 are gone with those
kernels; their logs are profiles/r2_call*/op_bench*.log.)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from sam_audio_amd import hip  # noqa: E402
from sam_audio_amd.config import preset_config  # noqa: E402
from tests import util  # noqa: E402

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
    print(f"{name:44s} {us:8.1f} us   {nbytes / us / 1e3:7.1f} GB/s  ({nbytes / 1e6:.0f} MB algorithmic)", flush=True)


qkv = torch.randn(M, 3 * D, device=dev).to(torch.bfloat16)
qw, kw = torch.rand(128, device=dev) + 0.5, torch.rand(128, device=dev) + 0.5
ang = torch.rand(10000, 64, device=dev)
cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
Q = torch.empty(B, H, Tp, 128, device=dev, dtype=torch.bfloat16)
K, Vt = torch.empty_like(Q), torch.empty(B, H, 128, Tp, device=dev, dtype=torch.bfloat16)
mask = torch.ones(B, T, dtype=torch.uint8, device=dev)
out = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
timeit("qkv_prep", lambda: hip.check(L.samaudio_op_qkv_prep(
    hip.ptr(qkv), hip.ptr(qw), hip.ptr(kw), hip.ptr(cos), hip.ptr(sin), hip.ptr(Q), hip.ptr(K), hip.ptr(Vt), hip.BF16, B, T,
    Tp, H, 1e-5, st())), 2 * M * 3 * D * 2)
timeit("self_attention", lambda: hip.check(L.samaudio_op_self_attention(
    hip.ptr(Q), hip.ptr(K), hip.ptr(Vt), hip.ptr(mask), hip.ptr(out), hip.BF16, B, T, Tp, H, st())), 4 * M * D * 2)
x = torch.randn(M, D, device=dev)
w = torch.rand(D, device=dev)
tab = torch.randn(6, D, device=dev)
t0 = torch.randn(1, 6 * D, device=dev)
xn = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
timeit("rmsnorm_mod (5 operand vectors per row)", lambda: hip.check(L.samaudio_op_rmsnorm_mod(
    hip.ptr(x), hip.ptr(w), C.c_void_p(tab[0].data_ptr()), C.c_void_p(tab[1].data_ptr()), hip.ptr(t0), 0, 0, D,
    hip.ptr(xn), hip.BF16, M, D, T, 1e-5, st())), M * D * 6)
q = torch.randn(M, D, device=dev).to(torch.bfloat16)
kv = torch.randn(B * Lt, 2 * D, device=dev).to(torch.bfloat16)
tmask = torch.ones(B, Lt, dtype=torch.uint8, device=dev)
timeit("cross_attention (unfolded, + k headnorm)", lambda: hip.check(L.samaudio_op_cross_attention(
    hip.ptr(q), hip.ptr(qw), hip.ptr(kv), hip.ptr(kw), hip.ptr(tmask), hip.ptr(out), hip.BF16, B, T, Lt, H, 1e-5, st())),
    2 * M * D * 2)
wo = (torch.randn(D, D, device=dev) / D ** 0.5).to(torch.bfloat16)
KP = (H * 8 + 63) // 64 * 64
ut = torch.zeros(B, D, KP, device=dev, dtype=torch.bfloat16)
kv_all = torch.randn(B * Lt, 22 * 2 * D, device=dev).to(torch.bfloat16)   # the engine's layout: all 22 layers' K | V per row
timeit("cross_attn_fold (kv_all layout)", lambda: hip.check(L.samaudio_op_cross_attn_fold(
    hip.ptr(wo), C.c_void_p(kv_all.data_ptr() + 5 * 2 * D * 2), 22 * 2 * D, hip.ptr(ut), KP, B, Lt, 8, H, st())),
    B * D * KP * 2 + D * D * 2)

# one DAC residual unit (k7 dilated conv -> Snake -> k1 conv + fp32 residual): what the tile policy launches for it
for Cc, Tc, items in ((64, 480000, 8), (96, 480000, 8), (128, 240000, 8), (192, 240000, 8)):
    geom = ((Tc + 80) * Cc, Cc, 40 * Cc)
    K7, K1 = (7 * Cc + 63) // 64 * 64, (Cc + 63) // 64 * 64
    xa = torch.randn(items, Tc + 80, Cc, device=dev).to(torch.bfloat16)
    mid, oc = torch.zeros_like(xa), torch.zeros_like(xa)
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
        for pp in (p7, p1):
            hip.check(L.samaudio_op_gemm(C.byref(pp), C.sizeof(pp), hip.BF16, st()))
    timeit(f"residual unit C={Cc} dil 3 [two launches]", two, unit_bytes, iters=5)
    L.samaudio_debug_set_flag(19, 2)
    timeit(f"residual unit C={Cc} dil 3 [fused, ring kernel]", lambda: util.resunit(p7, p1), unit_bytes, iters=5)
    L.samaudio_debug_set_flag(19, 0)
    if Cc <= 128:
        timeit(f"residual unit C={Cc} dil 3 [fused, weight-stationary]", lambda: util.resunit(p7, p1), unit_bytes, iters=5)
