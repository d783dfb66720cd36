"""Helpers shared by the GPU parity tests: thin ctypes callers of the per-kernel C-ABI hooks."""
import ctypes as C

import torch

from sam_audio_amd import hip

PREC = {"fp32": hip.F32, "bf16": hip.BF16, "fp16": hip.BF16, "mixed": hip.BF16}   # fp16 / mixed: libsamaudio_hip_f16.so
ACT_DT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16, "mixed": torch.float16}


def stream():
    return hip.current_stream_ptr()


def as_act(x: torch.Tensor, prec: str, dev) -> torch.Tensor:
    return x.to(dev, ACT_DT[prec]).contiguous()


def rounded(x: torch.Tensor, prec: str) -> torch.Tensor:
    """What the kernel actually sees: inputs rounded to the operand dtype, as fp32 on the CPU."""
    return x.to(ACT_DT[prec]).float()


def gemm_params(A, W, M, N, K, *, nbatch=1, a_off=0, a_bstride=0, lda=None, kc=None, tap_stride=0,
                bias=None, chan_mod=0, swiglu=0, gate_tab=None, gate=None, gate_ld=0, rows_per_gate=1, alpha=1.0,
                res=None, res_geom=(0, 0, 0), out_f32=None, f32_geom=(0, 0, 0), out_act=None, act_geom=(0, 0, 0),
                act=hip.ACT_NONE, f32_act=0, act_alpha=None, c_lo=0, c_hi=0, c_ld_rel=0, w_bstride=0, raster_gm=0, flags=0,
                prefetch=None):
    p = hip.GemmParams()
    p.A, p.W = A.data_ptr(), W.data_ptr()
    p.a_off, p.a_bstride, p.lda, p.tap_stride = a_off, a_bstride, (K if lda is None else lda), tap_stride
    p.kc = K if kc is None else kc
    p.M, p.N, p.K, p.nbatch = M, N, K, nbatch
    p.bias = 0 if bias is None else bias.data_ptr()
    p.chan_mod, p.swiglu = chan_mod, swiglu
    p.gate_tab = 0 if gate_tab is None else gate_tab.data_ptr()
    p.gate = 0 if gate is None else gate.data_ptr()
    p.gate_ld, p.rows_per_gate, p.alpha = gate_ld, rows_per_gate, alpha
    p.res = 0 if res is None else res.data_ptr()
    p.res_bstride, p.res_ld, p.res_off = res_geom
    p.out_f32 = 0 if out_f32 is None else out_f32.data_ptr()
    p.f32_bstride, p.f32_ld, p.f32_off = f32_geom
    p.out_act = 0 if out_act is None else out_act.data_ptr()
    p.act_bstride, p.act_ld, p.act_off = act_geom
    p.act, p.f32_act = act, f32_act
    p.act_alpha = 0 if act_alpha is None else act_alpha.data_ptr()
    p.c_lo, p.c_hi, p.c_ld_rel = c_lo, c_hi, c_ld_rel
    p.w_bstride, p.raster_gm = w_bstride, raster_gm
    p.flags = flags   # e.g. 2048: W is K-tile-major (common.h GEMM_FLAG_W_KTM)
    if prefetch is not None:   # bytes the launch's idle workgroups touch (the next launch's weights)
        p.pf_ptr, p.pf_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    return p


def gemm(prec: str, A, W, M, N, K, **kw):
    p = gemm_params(A, W, M, N, K, **kw)
    hip.check(hip.lib(hip.operands_for(prec)).samaudio_op_gemm(C.byref(p), C.sizeof(p), PREC[prec], stream()))


def resunit(p7, p1, prec="bf16"):
    """one DAC residual unit (k7 launch p7 + k1 launch p1 on its output) through the fused kernel"""
    hip.check(hip.lib(hip.operands_for(prec)).samaudio_op_resunit(C.byref(p7), C.byref(p1), C.sizeof(p7), stream()))


def report(name, got, want, tol):
    err = (got.float().cpu() - want).abs().max().item()
    scale = want.abs().max().item()
    print(f"{name}: max-abs err {err:.3e} (ref max {scale:.3f}, tol {tol:.1e})")
    assert err <= tol, f"{name}: {err} > {tol}"
    return err
