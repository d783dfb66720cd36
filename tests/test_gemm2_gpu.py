"""Parity of the 16-bit GEMM kernels of sam_audio_amd/csrc/gemm2.hip (32x32x16-MFMA ring tiles, conv7h, fused residual
units) and gemm8.hip (16x16x32-MFMA 8-phase 256x256 kernel and its 128x128 tile) through the C ABI, every tile variant the
policy can select forced in turn.

Checker: plain PyTorch fp32 on the CPU on the same bf16-rounded operands.  Tolerances: products of bf16 values are
exact in fp32 and the accumulation is fp32, so fp32 outputs agree to summation-order noise (<= 5e-4 at K <= 1344
on O(1) data); bf16 outputs add half a bf16 ulp (2^-9 relative to |value| <= 8 -> 3.2e-2).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

from sam_audio_amd import hip
from tests import util

pytestmark = pytest.mark.gpu
# gemm.hip gemm_variant numbering: 22 = gemm8 (8-phase 256x256), 27 = gemm8s (its 128x128 tile); 32x32x16 family: 25 / 26 =
# 128x128 / 64x128 (BK 64), 28 = 256x64, 29 / 32 / 33 / 34 = BK-32 multi-workgroup tiles; 35 = conv7h (its own tests).
VARIANTS = [22, 25, 26, 27, 28, 29, 32, 33, 34]


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(autouse=True)
def _restore_variant():
    yield
    hip.lib().samaudio_debug_force_gemm_variant(-1)
    hip.lib().samaudio_debug_set_flag(27, 0)


@pytest.mark.parametrize("variant", VARIANTS)
@pytest.mark.parametrize("M,N,K", [(300, 640, 192), (517, 1152, 320), (1000, 384, 1024), (250, 2816, 256), (64, 96, 704),
                                   (700, 192, 1344)])
def test_plain_tails(gpu, variant, M, N, K):
    hip.lib().samaudio_debug_force_gemm_variant(variant)
    A, W = _mk((M, K), 1), _mk((N, K), 2, 1 / math.sqrt(K))
    out = torch.full((M, N), float("nan"), device=gpu)
    util.gemm("bf16", util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), M, N, K, out_f32=out, f32_geom=(0, N, 0))
    want = util.rounded(A, "bf16") @ util.rounded(W, "bf16").T
    util.report(f"gemm2 v{variant} {M}x{N}x{K}", out, want, 5e-4)


@pytest.mark.parametrize("variant", VARIANTS)
def test_asymmetric_identity(gpu, variant):
    """A = I with an asymmetric W catches transposed / permuted output fragments exactly."""
    hip.lib().samaudio_debug_force_gemm_variant(variant)
    n = 512
    A = torch.eye(n)
    W = ((torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) / 8.0).to(torch.bfloat16).float()
    out = torch.empty(n, n, device=gpu)
    util.gemm("bf16", util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), n, n, n, out_f32=out, f32_geom=(0, n, 0))
    assert torch.equal(out.cpu(), W.T.contiguous())


@pytest.mark.parametrize("variant", VARIANTS)
def test_gate_residual_dual_output_and_swiglu(gpu, variant):
    hip.lib().samaudio_debug_force_gemm_variant(variant)
    B, T, N, K = 3, 70, 256, 320
    M = B * T
    A, W = _mk((M, K), 3), _mk((N, K), 4, 1 / math.sqrt(K))
    bias, tab, gate, res = _mk((N,), 5), _mk((N,), 6), _mk((B, 2 * N), 7), _mk((M, N), 8)
    out = torch.empty(M, N, device=gpu)
    out_act = torch.empty(M, N, device=gpu, dtype=torch.bfloat16)
    gate_d = gate.to(gpu)
    keep = [bias.to(gpu), tab.to(gpu), res.to(gpu)]
    util.gemm("bf16", util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), M, N, K, bias=keep[0], gate_tab=keep[1],
              gate=gate_d[:, N:], gate_ld=2 * N, rows_per_gate=T, alpha=0.5, res=keep[2], res_geom=(0, N, 0),
              out_f32=out, f32_geom=(0, N, 0), out_act=out_act, act_geom=(0, N, 0), act=hip.ACT_SILU)
    base = util.rounded(A, "bf16") @ util.rounded(W, "bf16").T + bias
    want = base * (tab[None] + gate[:, N:].repeat_interleave(T, 0)) * 0.5 + res
    util.report(f"gated v{variant} f32", out, want, 5e-4)
    util.report(f"gated v{variant} act", out_act, F.silu(want), 3.2e-2)
    # swiglu: weights interleaved in 16-row blocks (w1 block, w3 block, ...)
    Fh = 192
    w1, w3 = _mk((Fh, K), 9, 1 / math.sqrt(K)), _mk((Fh, K), 10, 1 / math.sqrt(K))
    W13 = torch.stack([w1.view(Fh // 16, 16, K), w3.view(Fh // 16, 16, K)], 1).reshape(2 * Fh, K)
    u = torch.empty(M, Fh, device=gpu, dtype=torch.bfloat16)
    util.gemm("bf16", util.as_act(A, "bf16", gpu), util.as_act(W13, "bf16", gpu), M, 2 * Fh, K, swiglu=1, out_act=u,
              act_geom=(0, Fh, 0))
    Ar = util.rounded(A, "bf16")
    want_u = F.silu(Ar @ util.rounded(w1, "bf16").T) * (Ar @ util.rounded(w3, "bf16").T)
    util.report(f"swiglu v{variant}", u, want_u, 3.2e-2)


@pytest.mark.parametrize("variant", VARIANTS)
def test_conv_forms(gpu, variant):
    """The codec's implicit-convolution forms on the 256-row kernels: dilated k7 conv with snake epilogue into a
    halo-padded buffer, and a stride-4 transposed conv (phase-major columns, chan_mod bias, output window mask)."""
    hip.lib().samaudio_debug_force_gemm_variant(variant)
    items, T, C, dil, halo = 2, 300, 128, 3, 40
    x = _mk((items, C, T), 11)
    w = _mk((C, C, 7), 12, 1 / math.sqrt(7 * C))
    bias, alpha = _mk((C,), 13, 0.1), (_mk((C,), 14, 0.2) + 1).clamp(0.3, 2)
    xb = torch.zeros(items, T + 2 * halo, C)
    xb[:, halo:halo + T] = x.transpose(1, 2)
    Wm = w.permute(0, 2, 1).reshape(C, 7 * C)  # [Cout][tap][Cin]
    Kp = 7 * C
    out_act = torch.zeros(items, T + 2 * halo, C, device=gpu, dtype=torch.bfloat16)
    keep = [bias.to(gpu), alpha.to(gpu)]
    util.gemm("bf16", util.as_act(xb, "bf16", gpu), util.as_act(Wm, "bf16", gpu), T, C, Kp, nbatch=items,
              a_off=(halo - 3 * dil) * C, a_bstride=(T + 2 * halo) * C, lda=C, kc=C, tap_stride=dil * C, bias=keep[0],
              out_act=out_act, act_geom=((T + 2 * halo) * C, C, halo * C), act=hip.ACT_SNAKE, act_alpha=keep[1])
    y = F.conv1d(util.rounded(x, "bf16"), util.rounded(w, "bf16"), bias, dilation=dil, padding=3 * dil)
    a = alpha[None, :, None]
    want = (y + torch.sin(a * y) ** 2 / (a + 1e-9)).transpose(1, 2)
    util.report(f"dilated conv+snake v{variant}", out_act[:, halo:halo + T], want, 4e-2)
    assert float(out_act[:, :halo].abs().max()) == 0 and float(out_act[:, halo + T:].abs().max()) == 0

    # ConvTranspose1d(k = 2s, stride s, padding s/2): rows q, columns (phase r, channel c)
    s, Cin, Co, Tin = 4, 128, 64, 150
    pad = s // 2
    xt = _mk((items, Cin, Tin), 15)
    wt = _mk((Cin, Co, 2 * s), 16, 1 / math.sqrt(2 * Cin))
    bt = _mk((Co,), 17, 0.1)
    xin = torch.zeros(items, Tin + 2 * halo, Cin)
    xin[:, halo:halo + Tin] = xt.transpose(1, 2)
    # W[(r, co)][ (x[q-1] | x[q]) ] = (w[:, co, r + s] | w[:, co, r])
    Wt = torch.cat([wt[:, :, s:].permute(2, 1, 0), wt[:, :, :s].permute(2, 1, 0)], dim=2).reshape(s * Co, 2 * Cin)
    Tout = Tin * s
    raw = torch.zeros(items, Tout + 2 * halo, Co, device=gpu)
    kb = bt.to(gpu)
    util.gemm("bf16", util.as_act(xin, "bf16", gpu), util.as_act(Wt, "bf16", gpu), Tin + 1, s * Co, 2 * Cin, nbatch=items,
              a_off=(halo - 1) * Cin, a_bstride=(Tin + 2 * halo) * Cin, lda=Cin, bias=kb, chan_mod=Co, out_f32=raw,
              f32_geom=((Tout + 2 * halo) * Co, s * Co, (halo - pad) * Co), c_lo=pad * Co, c_hi=(Tout + pad) * Co,
              c_ld_rel=s * Co)
    want_t = F.conv_transpose1d(util.rounded(xt, "bf16"), util.rounded(wt, "bf16"), bt, stride=s, padding=pad).transpose(1, 2)
    util.report(f"convT v{variant}", raw[:, halo:halo + Tout], want_t, 5e-4)
    assert float(raw[:, :halo].abs().max()) == 0 and float(raw[:, halo + Tout:].abs().max()) == 0


def test_row_tile_variants_are_bitwise_identical(gpu):
    """The M-aware policy may pick 256-, 128- or 64-row tiles of the 32x32x16-MFMA family for the SAME (N, K) depending on
    how many rows a launch has (sam_audio_amd/csrc/gemm.hip gemm_variant).  Batch sharding stays bitwise invariant
     only if every one of them accumulates an output element in the same order: same MFMA shape,
    K walked slab by slab.  Gated-residual epilogue, fp32 + bf16 outputs, ragged M."""
    B, T, N, K = 3, 90, 384, 448
    M = B * T
    A, W = _mk((M, K), 21), _mk((N, K), 22, 1 / math.sqrt(K))
    tab, gate, res = _mk((N,), 23), _mk((B, N), 24), _mk((M, N), 25)
    keep = [util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), tab.to(gpu), gate.to(gpu), res.to(gpu)]
    outs = {}
    for variant in (25, 26, 28, 29, 32, 33, 34):
        hip.lib().samaudio_debug_force_gemm_variant(variant)
        out = torch.full((M, N), float("nan"), device=gpu)
        out_act = torch.zeros(M, N, device=gpu, dtype=torch.bfloat16)
        util.gemm("bf16", keep[0], keep[1], M, N, K, gate_tab=keep[2], gate=keep[3], gate_ld=N, rows_per_gate=T, res=keep[4],
                  res_geom=(0, N, 0), out_f32=out, f32_geom=(0, N, 0), out_act=out_act, act_geom=(0, N, 0))
        outs[variant] = (out.cpu(), out_act.cpu())
    for variant, (o, a) in outs.items():
        assert torch.equal(o, outs[25][0]) and torch.equal(a.view(torch.int16), outs[25][1].view(torch.int16)), variant


@pytest.mark.parametrize("M,N,K,swiglu", [(270, 384, 448, 0), (700, 2816, 256, 0), (333, 512, 320, 1)])
def test_8phase_family_tiles_are_bitwise_identical(gpu, M, N, K, swiglu):
    """Same property for the 16x16x32-MFMA family that carries every N >= 2048 GEMM: the 256x256 8-phase kernel (22) and
    its 128x128 tile gemm8s (27, picked for launches with few rows) must produce identical bits - gated-residual
    epilogue with fp32 + bf16 outputs, and the SwiGLU epilogue; ragged M and N tails."""
    B = 1
    A, W = _mk((M, K), 31), _mk((N, K), 32, 1 / math.sqrt(K))
    n_out = N // 2 if swiglu else N
    tab, gate, res = _mk((n_out,), 33), _mk((B, n_out), 34), _mk((M, n_out), 35)
    keep = [util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), tab.to(gpu), gate.to(gpu), res.to(gpu)]
    outs = {}
    for variant in (22, 27):
        hip.lib().samaudio_debug_force_gemm_variant(variant)
        out_act = torch.zeros(M, n_out, device=gpu, dtype=torch.bfloat16)
        if swiglu:
            util.gemm("bf16", keep[0], keep[1], M, N, K, swiglu=1, out_act=out_act, act_geom=(0, n_out, 0))
            outs[variant] = (out_act.cpu(), out_act.cpu())
        else:
            out = torch.full((M, N), float("nan"), device=gpu)
            util.gemm("bf16", keep[0], keep[1], M, N, K, gate_tab=keep[2], gate=keep[3], gate_ld=N, rows_per_gate=M,
                      res=keep[4], res_geom=(0, N, 0), out_f32=out, f32_geom=(0, N, 0), out_act=out_act,
                      act_geom=(0, N, 0))
            outs[variant] = (out.cpu(), out_act.cpu())
    o22, a22 = outs[22]
    o27, a27 = outs[27]
    assert torch.isfinite(o22.float()).all()
    assert torch.equal(o22.view(torch.int16) if swiglu else o22, o27.view(torch.int16) if swiglu else o27)
    assert torch.equal(a22.view(torch.int16), a27.view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(200, 256, 64), (130, 384, 128), (333, 512, 192), (270, 2816, 448), (250, 640, 512),
                                   (130, 256, 704), (150, 384, 832), (300, 512, 1408)])
def test_gemm8s_pipelined_form_is_bitwise_identical(gpu, M, N, K):
    """gemm8s runs launches of <= 256 workgroups in its pipelined form (3-stage ring, the fragments of K-tile t+1 read
    underneath the MFMAs of K-tile t; debug flag 21 = the plain double-buffered form).  Same arithmetic: identical bits
    for 1 .. 8 K-tiles (odd and even counts, shorter than the ring) and 11 / 13 / 22, against the plain form and against the 256x256
    kernel; gated-residual epilogue, fp32 + bf16 outputs."""
    A, W = _mk((M, K), 51), _mk((N, K), 52, 1 / math.sqrt(K))
    tab, gate, res = _mk((N,), 53), _mk((1, N), 54), _mk((M, N), 55)
    keep = [util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), tab.to(gpu), gate.to(gpu), res.to(gpu)]
    outs = {}
    try:
        # flag 27: the pipelined form's wave roles - 1 = none (4 waves request and multiply), 2 / 3 = 4 requesting waves beside
        # 4 multiplying ones (the latter issuing 0 / 2 of their loads themselves); 0 = the shipped choice
        for name, variant, flag, roles in (("pipelined", 27, 0, 0), ("no roles", 27, 0, 1), ("roles 0", 27, 0, 2),
                                           ("roles 2", 27, 0, 3), ("plain", 27, 1, 0), ("8phase", 22, 0, 0)):
            hip.lib().samaudio_debug_force_gemm_variant(variant)
            hip.lib().samaudio_debug_set_flag(21, flag)
            hip.lib().samaudio_debug_set_flag(27, roles)
            out = torch.full((M, N), float("nan"), device=gpu)
            out_act = torch.zeros(M, N, device=gpu, dtype=torch.bfloat16)
            util.gemm("bf16", keep[0], keep[1], M, N, K, gate_tab=keep[2], gate=keep[3], gate_ld=N, rows_per_gate=M,
                      res=keep[4], res_geom=(0, N, 0), out_f32=out, f32_geom=(0, N, 0), out_act=out_act, act_geom=(0, N, 0))
            outs[name] = (out.cpu(), out_act.cpu())
    finally:
        hip.lib().samaudio_debug_set_flag(21, 0)
        hip.lib().samaudio_debug_set_flag(27, 0)
        hip.lib().samaudio_debug_force_gemm_variant(-1)
    want = (util.rounded(A, "bf16") @ util.rounded(W, "bf16").T) * (tab[None] + gate) + res
    util.report(f"gemm8s pipelined {M}x{N}x{K}", outs["pipelined"][0], want, 5e-4)
    for other in ("no roles", "roles 0", "roles 2", "plain", "8phase"):
        assert torch.equal(outs["pipelined"][0], outs[other][0])
        assert torch.equal(outs["pipelined"][1].view(torch.int16), outs[other][1].view(torch.int16))


@pytest.mark.parametrize("M,N,K", [(270, 2816, 448), (333, 512, 192), (130, 320, 128), (700, 1024, 704)])
@pytest.mark.parametrize("kind", ["gated", "swiglu"])
def test_ktm_weights_and_prefetch_workgroups_are_bitwise_invisible(gpu, M, N, K, kind):
    """GemmParams.flags bit 11: W stored K-tile-major [K/64][N][64] (weights.ktm_layout) instead of [N][K] - the 8-phase family
    reads either layout into the same LDS image.  GemmParams.pf_ptr: launches of fewer than 256 workgroups are padded with
    workgroups that only touch the next launch's weights.  Every form of the family (256x256; 128x128 plain / pipelined without
    and with requesting waves), both layouts, with and without the prefetch: identical bits; the row-major result is checked
    against fp32 torch."""
    from sam_audio_amd.weights import ktm_layout, ktm_to_rows
    A, W = _mk((M, K), 91), _mk((N, K), 92, 1 / math.sqrt(K))
    Wb = util.as_act(W, "bf16", gpu)
    Wk = ktm_layout(Wb)
    assert Wk.shape == (K // 64, N, 64) and torch.equal(ktm_to_rows(Wk), Wb)
    nxt = util.as_act(_mk((333, 77), 93), "bf16", gpu)    # "the next launch's weights": 51 282 bytes, not a multiple of 128
    tab, gate, res = _mk((N,), 94), _mk((1, N), 95), _mk((M, N), 96)
    keep = [util.as_act(A, "bf16", gpu), tab.to(gpu), gate.to(gpu), res.to(gpu)]
    n_out = N // 2 if kind == "swiglu" else N
    outs = {}
    try:
        for variant, flag21, roles in ((22, 0, 0), (27, 0, 1), (27, 0, 2), (27, 0, 3), (27, 1, 0)):
            for layout in ("rows", "ktm"):
                for pf in (False, True):
                    hip.lib().samaudio_debug_force_gemm_variant(variant)
                    hip.lib().samaudio_debug_set_flag(21, flag21)
                    hip.lib().samaudio_debug_set_flag(27, roles)
                    out = torch.full((M, n_out), float("nan"), device=gpu)
                    out_act = torch.zeros(M, n_out, device=gpu, dtype=torch.bfloat16)
                    kw = dict(out_act=out_act, act_geom=(0, n_out, 0), flags=2048 if layout == "ktm" else 0,
                              prefetch=nxt if pf else None)
                    if kind == "swiglu":
                        kw.update(swiglu=1)
                    else:
                        kw.update(gate_tab=keep[1], gate=keep[2], gate_ld=N, rows_per_gate=M, res=keep[3], res_geom=(0, N, 0),
                                  out_f32=out, f32_geom=(0, N, 0))
                    util.gemm("bf16", keep[0], Wk if layout == "ktm" else Wb, M, N, K, **kw)
                    outs[(variant, flag21, roles, layout, pf)] = (out.cpu(), out_act.cpu())
    finally:
        hip.lib().samaudio_debug_set_flag(21, 0)
        hip.lib().samaudio_debug_set_flag(27, 0)
        hip.lib().samaudio_debug_force_gemm_variant(-1)
    first = outs[(22, 0, 0, "rows", False)]
    if kind == "gated":
        want = (util.rounded(A, "bf16") @ util.rounded(W, "bf16").T) * (tab[None] + gate) + res
        util.report(f"gemm8 rows {M}x{N}x{K}", first[0], want, 5e-4)
    else:
        assert torch.isfinite(first[1].float()).all() and float(first[1].float().abs().max()) > 0
    for key, (o, a) in outs.items():
        if kind == "gated":
            assert torch.equal(first[0], o), key
        assert torch.equal(first[1].view(torch.int16), a.view(torch.int16)), key


def test_ktm_weights_are_refused_outside_the_8phase_family(gpu):
    """Only gemm8 / gemm8s address K-tile-major weights: any other tile choice (here a forced one) must fail loudly."""
    from sam_audio_amd.weights import ktm_layout
    M, N, K = 300, 512, 192
    A, W = util.as_act(_mk((M, K), 97), "bf16", gpu), ktm_layout(util.as_act(_mk((N, K), 98), "bf16", gpu))
    out = torch.zeros(M, N, device=gpu)
    hip.lib().samaudio_debug_force_gemm_variant(25)
    with pytest.raises(AssertionError, match="K-tile-major"):
        util.gemm("bf16", A, W, M, N, K, out_f32=out, f32_geom=(0, N, 0), flags=2048)


@pytest.mark.parametrize("kind", ["conv", "plain16", "swiglu"])
def test_gemm8s_wave_roles_are_bitwise_invisible(gpu, kind):
    """The pipelined form with requesting waves (debug flag 27 = 2 / 3) against the one without (1), on the launches the first
    test does not reach: an implicit dilated k7 convolution with the Snake epilogue (per-lane source pointers, the tap walk
    advanced by whichever wave requests a row), a 16-bit-only output (register epilogue: the requesting waves leave without
    the epilogue barrier) and SwiGLU."""
    outs = {}
    try:
        hip.lib().samaudio_debug_force_gemm_variant(27)
        if kind == "conv":
            items, T, C, dil, halo = 2, 300, 256, 3, 40
            x, w = _mk((items, C, T), 81), _mk((C, C, 7), 82, 1 / math.sqrt(7 * C))
            bias, alpha = _mk((C,), 83, 0.1), (_mk((C,), 84, 0.2) + 1).clamp(0.3, 2)
            xb = torch.zeros(items, T + 2 * halo, C)
            xb[:, halo:halo + T] = x.transpose(1, 2)
            keep = [util.as_act(xb, "bf16", gpu), util.as_act(w.permute(0, 2, 1).reshape(C, 7 * C), "bf16", gpu), bias.to(gpu), alpha.to(gpu)]
        else:
            M, N, K = 300, 768, 448
            A, W = _mk((M, K), 85), _mk((N, K), 86, 1 / math.sqrt(K))
            keep = [util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu)]
        for roles in (1, 2, 3):
            hip.lib().samaudio_debug_set_flag(27, roles)
            if kind == "conv":
                out = torch.zeros(items, T + 2 * halo, C, device=gpu, dtype=torch.bfloat16)
                util.gemm("bf16", keep[0], keep[1], T, C, 7 * C, nbatch=items, a_off=(halo - 3 * dil) * C,
                          a_bstride=(T + 2 * halo) * C, lda=C, kc=C, tap_stride=dil * C, bias=keep[2], out_act=out,
                          act_geom=((T + 2 * halo) * C, C, halo * C), act=hip.ACT_SNAKE, act_alpha=keep[3])
            else:
                n_out = N // 2 if kind == "swiglu" else N
                out = torch.zeros(M, n_out, device=gpu, dtype=torch.bfloat16)
                util.gemm("bf16", keep[0], keep[1], M, N, K, out_act=out, act_geom=(0, n_out, 0), swiglu=int(kind == "swiglu"))
            outs[roles] = out.cpu()
    finally:
        hip.lib().samaudio_debug_set_flag(27, 0)
        hip.lib().samaudio_debug_force_gemm_variant(-1)
    assert torch.isfinite(outs[1].float()).all() and float(outs[1].float().abs().max()) > 0
    for roles in (2, 3):
        assert torch.equal(outs[1].view(torch.int16), outs[roles].view(torch.int16)), f"flag 27 = {roles}"
    if kind == "plain16":
        util.report("gemm8s roles, 16-bit output", outs[2], util.rounded(A, "bf16") @ util.rounded(W, "bf16").T, 3.2e-2)


@pytest.mark.parametrize("M,N,K,nbatch", [(4352, 4096, 128, 1), (300, 3500, 192, 10), (2000, 8448, 128, 1)])
def test_tail_split_is_bitwise_invisible(gpu, M, N, K, nbatch):
    """8-phase launches whose last round of 256x256 tiles is mostly empty run as two kernels (gemm.hip gemm_tail_split:
    whole rounds on gemm8, the rest as 128x128 quadrants on gemm8s).  272 / 280 / 264 tiles here -> 256 + 16 / 24 / 8 (the last
    = the qkv GEMM of 8 clips per GPU: the smallest tail that is split off since round 5); ragged M and N
    put quadrants partly and wholly outside the problem.  The automatic (split) result must equal the forced single-kernel
    one bit for bit - gated residual epilogue, fp32 + bf16 outputs, per-batch strides."""
    A, W = _mk((nbatch, M, K), 41), _mk((N, K), 42, 1 / math.sqrt(K))
    tab, gate, res = _mk((N,), 43), _mk((nbatch, N), 44), _mk((nbatch, M, N), 45)
    keep = [util.as_act(A, "bf16", gpu), util.as_act(W, "bf16", gpu), tab.to(gpu), gate.to(gpu), res.to(gpu)]
    outs = {}
    for variant in (-1, 22):
        hip.lib().samaudio_debug_force_gemm_variant(variant)
        out = torch.full((nbatch, M, N), float("nan"), device=gpu)
        out_act = torch.zeros(nbatch, M, N, device=gpu, dtype=torch.bfloat16)
        util.gemm("bf16", keep[0], keep[1], M, N, K, nbatch=nbatch, a_bstride=M * K, gate_tab=keep[2], gate=keep[3],
                  gate_ld=N, rows_per_gate=M, res=keep[4], res_geom=(M * N, N, 0), out_f32=out, f32_geom=(M * N, N, 0),
                  out_act=out_act, act_geom=(M * N, N, 0))
        outs[variant] = (out.cpu(), out_act.cpu())
    want = (util.rounded(A, "bf16") @ util.rounded(W, "bf16").T) * (tab[None, None] + gate[:, None, :]) + res
    util.report(f"tail split {M}x{N}x{K}x{nbatch}", outs[-1][0], want, 5e-4)
    assert torch.equal(outs[-1][0], outs[22][0])
    assert torch.equal(outs[-1][1].view(torch.int16), outs[22][1].view(torch.int16))


def test_sixty_four_channel_tiles_are_bitwise_identical(gpu):
    """N = 64 outputs (first DAC encoder stage): the 256x64 tile (28) and its few-rows fallback, the 128x64 BK-32 tile (32),
    must agree bit for bit - the number of waveforms per codec pass (hence which of the two runs) depends on the caller's
    workspace.  Dilated k7 convolution form with bias + Snake, and the k1 form with an in-place fp32 residual."""
    items, T, C, dil, halo = 3, 700, 64, 3, 40
    x = _mk((items, C, T), 51)
    xb = torch.zeros(items, T + 2 * halo, C)
    xb[:, halo:halo + T] = x.transpose(1, 2)
    w7 = _mk((C, 7 * C), 52, 1 / math.sqrt(7 * C))
    w1 = _mk((C, C), 53, 1 / math.sqrt(C))
    bias, alpha = _mk((C,), 54, 0.1), (_mk((C,), 55, 0.2) + 1).clamp(0.3, 2)
    raw0 = _mk((items, T + 2 * halo, C), 56)
    keep = [util.as_act(xb, "bf16", gpu), util.as_act(w7, "bf16", gpu), util.as_act(w1, "bf16", gpu), bias.to(gpu), alpha.to(gpu)]
    outs = {}
    for variant in (28, 32):
        hip.lib().samaudio_debug_force_gemm_variant(variant)
        tmp = torch.zeros(items, T + 2 * halo, C, device=gpu, dtype=torch.bfloat16)
        util.gemm("bf16", keep[0], keep[1], T, C, 7 * C, nbatch=items, a_off=(halo - 3 * dil) * C, a_bstride=(T + 2 * halo) * C,
                  lda=C, kc=C, tap_stride=dil * C, bias=keep[3], out_act=tmp, act_geom=((T + 2 * halo) * C, C, halo * C),
                  act=hip.ACT_SNAKE, act_alpha=keep[4])
        raw = raw0.to(gpu)
        act = torch.zeros_like(tmp)
        util.gemm("bf16", tmp, keep[2], T, C, C, nbatch=items, a_off=halo * C, a_bstride=(T + 2 * halo) * C, lda=C, bias=keep[3],
                  res=raw, res_geom=((T + 2 * halo) * C, C, halo * C), out_f32=raw, f32_geom=((T + 2 * halo) * C, C, halo * C),
                  out_act=act, act_geom=((T + 2 * halo) * C, C, halo * C), act=hip.ACT_SNAKE, act_alpha=keep[4])
        outs[variant] = (tmp.cpu(), raw.cpu(), act.cpu())
    for a, b_ in zip(outs[28], outs[32]):
        assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b_.view(torch.int16) if b_.dtype == torch.bfloat16 else b_)
    y = torch.nn.functional.conv1d(util.rounded(x, "bf16"), util.rounded(w7, "bf16").view(C, 7, C).permute(0, 2, 1), bias,
                                   dilation=dil, padding=3 * dil)
    a_ = alpha[None, :, None]
    want = (y + torch.sin(a_ * y) ** 2 / (a_ + 1e-9)).transpose(1, 2)
    util.report("k7 C=64 conv + snake (256x64 tile)", outs[28][0][:, halo:halo + T], want, 4e-2)


@pytest.mark.parametrize("C,dil,T,items", [(64, 1, 700, 3), (96, 3, 530, 2), (96, 9, 300, 3), (128, 9, 520, 2), (192, 3, 300, 3),
                                           (192, 1, 130, 2)])
def test_conv7h_is_bitwise_the_implicit_gemm(gpu, C, dil, T, items):
    """conv7h (variant 35: k7 'same' convolution with the activation halo tile resident in LDS, taps walked by shifting
    fragment rows) against the implicit GEMM of the same MFMA family (variant 25) on identical operands: identical bits -
    bias + Snake epilogue into a halo-padded buffer, every channel count / dilation the DAC stages use, M not a multiple
    of the row tile (the last tile's halo rows are clamped), several items; and both against torch's conv1d."""
    halo = 40
    x = _mk((items, C, T), 61)
    xb = torch.zeros(items, T + 2 * halo, C)
    xb[:, halo:halo + T] = x.transpose(1, 2)
    Kp = (7 * C + 63) // 64 * 64
    w = _mk((C, C, 7), 62, 1 / math.sqrt(7 * C))
    Wm = torch.zeros(C, Kp)
    Wm[:, : 7 * C] = w.permute(0, 2, 1).reshape(C, 7 * C)   # [Cout][tap][Cin]
    bias, alpha = _mk((C,), 63, 0.1), (_mk((C,), 64, 0.2) + 1).clamp(0.3, 2)
    keep = [util.as_act(xb, "bf16", gpu), util.as_act(Wm, "bf16", gpu), bias.to(gpu), alpha.to(gpu)]
    outs = {}
    for variant in (35, 25):
        hip.lib().samaudio_debug_force_gemm_variant(variant)
        out = torch.zeros(items, T + 2 * halo, C, device=gpu, dtype=torch.bfloat16)
        util.gemm("bf16", keep[0], keep[1], T, C, Kp, nbatch=items, a_off=(halo - 3 * dil) * C, a_bstride=(T + 2 * halo) * C,
                  lda=C, kc=C, tap_stride=dil * C, bias=keep[2], out_act=out, act_geom=((T + 2 * halo) * C, C, halo * C),
                  act=hip.ACT_SNAKE, act_alpha=keep[3])
        outs[variant] = out.cpu()
    assert torch.equal(outs[35].view(torch.int16), outs[25].view(torch.int16))
    assert float(outs[35][:, :halo].abs().max()) == 0 and float(outs[35][:, halo + T:].abs().max()) == 0
    y = F.conv1d(util.rounded(x, "bf16"), util.rounded(w, "bf16"), bias, dilation=dil, padding=3 * dil)
    a = alpha[None, :, None]
    want = (y + torch.sin(a * y) ** 2 / (a + 1e-9)).transpose(1, 2)
    util.report(f"conv7h C={C} dil={dil}", outs[35][:, halo:halo + T], want, 4e-2)


@pytest.mark.parametrize("C,dil,T,items", [(64, 1, 700, 3), (96, 3, 530, 2), (96, 9, 300, 3), (128, 9, 520, 2), (192, 3, 300, 3),
                                           (192, 1, 130, 2), (96, 1, 256, 1), (96, 1, 128, 2), (96, 1, 1100, 2),
                                           (64, 3, 128 * 90 + 17, 3)])
def test_fused_residual_unit_is_bitwise_the_two_launches(gpu, C, dil, T, items):
    """resunit (one DAC residual unit per launch: k7 convolution -> Snake -> bf16 intermediate kept in LDS -> k1 convolution
    + fp32 residual, fp32 stream and Snake'd bf16 copy out) against the two launches the engine otherwise issues, on identical
    operands: identical bits in both outputs, halo rows of the output activation untouched, the input activation untouched;
    every channel count / dilation the DAC stages use, M not a multiple of the row tile, several items.  Both fused kernels:
    the ring kernel (one tile per workgroup) and, for C <= 96, the weight-stationary persistent kernel that large launches
    use (debug flag 19: one tile per workgroup, and 3 workgroups walking all tiles; the last case has > 256 tiles = the
    full-grid tile order).  The two-launch form itself is checked against torch (conv1d + Snake + conv1d + residual)."""
    halo = 40
    x = _mk((items, C, T), 71)
    xb = torch.zeros(items, T + 2 * halo, C)
    xb[:, halo:halo + T] = x.transpose(1, 2)
    K7, K1 = (7 * C + 63) // 64 * 64, (C + 63) // 64 * 64
    w7 = _mk((C, C, 7), 72, 1 / math.sqrt(7 * C))
    W7 = torch.zeros(C, K7)
    W7[:, : 7 * C] = w7.permute(0, 2, 1).reshape(C, 7 * C)
    w1 = _mk((C, C), 73, 1 / math.sqrt(C))
    W1 = torch.zeros(C, K1)
    W1[:, :C] = w1
    b7, a7 = _mk((C,), 74, 0.1), (_mk((C,), 75, 0.2) + 1).clamp(0.3, 2)
    b1, a1 = _mk((C,), 76, 0.1), (_mk((C,), 77, 0.2) + 1).clamp(0.3, 2)
    raw0 = torch.zeros(items, T + 2 * halo, C)
    raw0[:, halo:halo + T] = _mk((items, T, C), 78)
    xin = util.as_act(xb, "bf16", gpu)
    keep = [xin, util.as_act(W7, "bf16", gpu), util.as_act(W1, "bf16", gpu), b7.to(gpu), a7.to(gpu), b1.to(gpu), a1.to(gpu)]
    geom = ((T + 2 * halo) * C, C, halo * C)

    def params(mid, raw, out):
        p7 = util.gemm_params(keep[0], keep[1], T, C, K7, nbatch=items, a_off=(halo - 3 * dil) * C, a_bstride=geom[0], lda=C,
                              kc=C, tap_stride=dil * C, bias=keep[3], out_act=mid, act_geom=geom, act=hip.ACT_SNAKE,
                              act_alpha=keep[4])
        p1 = util.gemm_params(mid, keep[2], T, C, K1, nbatch=items, a_off=halo * C, a_bstride=geom[0], lda=C, kc=C, tap_stride=C,
                              bias=keep[5], res=raw, res_geom=geom, out_f32=raw, f32_geom=geom, out_act=out, act_geom=geom,
                              act=hip.ACT_SNAKE, act_alpha=keep[6])
        return p7, p1

    import ctypes as CT
    res = {}
    forms = [(False, 2), (True, 2)] + ([(True, 1), (True, 3)] if C <= 96 else [])   # (fused, debug flag 19)
    try:
        for fused, ws in forms:
            hip.lib().samaudio_debug_set_flag(19, ws)
            mid = torch.zeros(items, T + 2 * halo, C, device=gpu, dtype=torch.bfloat16)
            out = torch.zeros(items, T + 2 * halo, C, device=gpu, dtype=torch.bfloat16)
            raw = raw0.to(gpu)
            p7, p1 = params(mid, raw, out)
            if fused:
                util.resunit(p7, p1)
                assert float(mid.float().abs().max()) == 0          # the intermediate never reaches memory
            else:
                for p in (p7, p1):
                    hip.check(hip.lib().samaudio_op_gemm(CT.byref(p), CT.sizeof(p), hip.BF16, util.stream()))
            res[(fused, ws)] = (raw.cpu(), out.cpu())
    finally:
        hip.lib().samaudio_debug_set_flag(19, 0)
    assert torch.equal(xin.cpu().view(torch.int16), util.as_act(xb, "bf16", "cpu").view(torch.int16))
    for form in forms[1:]:
        assert torch.equal(res[form][0].view(torch.int32), res[(False, 2)][0].view(torch.int32)), form
        assert torch.equal(res[form][1].view(torch.int16), res[(False, 2)][1].view(torch.int16)), form
        assert float(res[form][1][:, :halo].abs().max()) == 0 and float(res[form][1][:, halo + T:].abs().max()) == 0
    res = {True: res[forms[-1]], False: res[(False, 2)]}
    snake = lambda y, a: y + torch.sin(a[None, :, None] * y) ** 2 / (a[None, :, None] + 1e-9)   # noqa: E731
    y = util.rounded(snake(F.conv1d(util.rounded(x, "bf16"), util.rounded(w7, "bf16"), b7, dilation=dil, padding=3 * dil), a7), "bf16")
    want_raw = raw0[:, halo:halo + T] + (F.conv1d(y, util.rounded(w1, "bf16")[:, :, None], b1)).transpose(1, 2)
    util.report(f"resunit C={C} dil={dil} raw", res[True][0][:, halo:halo + T], want_raw, 4e-2)
    util.report(f"resunit C={C} dil={dil} act", res[True][1][:, halo:halo + T], snake(want_raw.transpose(1, 2), a1).transpose(1, 2), 6e-2)
    # the output activation must not alias the input (neighbouring tiles read input halo rows): rejected, not raced
    p7, p1 = params(torch.zeros_like(xin), raw0.to(gpu), xin)
    assert hip.lib().samaudio_op_resunit(CT.byref(p7), CT.byref(p1), CT.sizeof(p7), util.stream()) == hip.ERR_ARG


@pytest.mark.parametrize("variant", [22, 27])
@pytest.mark.parametrize("kind", ["act", "f32", "gated_dual", "gated_f32", "swiglu", "bias_res_batched", "alpha_res"])
def test_linear_epilogue_is_bitwise_the_general_one(gpu, variant, kind):
    """gemm8.hip's lean epilogues of the DiT's Linears (bias | gate + table, residual, fp32 / 16-bit outputs, SwiGLU) -
    epilogue8_linear straight from the accumulator layout and epilogue8_rows through the wave's LDS area; debug flag 24: 0 =
    the shipped choice, 2 / 3 = one form for every eligible launch - against the general epilogue (flag 24 = 1) on the same
    launch: identical bits; ragged M (masked rows inside a tile), N with waves wholly outside the problem, batch strides;
    alpha != 1 is outside the lean contract and must simply agree."""
    M, N, K, nb = 333, 448, 192, 1
    kw = {}
    g = lambda *shape, seed: _mk(shape, seed).to(gpu)   # noqa: E731
    if kind == "bias_res_batched":
        M, N, K, nb = 250, 320, 128, 3
    A, W = _mk((nb, M, K), 61), _mk((N, K), 62, 1 / math.sqrt(K))
    n_out = N // 2 if kind == "swiglu" else N
    outs = {}
    keep = dict(A=util.as_act(A, "bf16", gpu), W=util.as_act(W, "bf16", gpu), tab=g(N, seed=63), gate=g(4, N, seed=64),
                res=g(nb, M, N, seed=65), bias=g(N, seed=66))
    try:
        for flag in (1, 0, 2, 3):
            hip.lib().samaudio_debug_force_gemm_variant(variant)
            hip.lib().samaudio_debug_set_flag(24, flag)
            o32 = torch.full((nb, M, N), float("nan"), device=gpu)
            o16 = torch.zeros(nb, M, n_out, device=gpu, dtype=torch.bfloat16)
            if kind == "act":
                kw = dict(out_act=o16, act_geom=(M * N, N, 0))
            elif kind == "f32":
                kw = dict(out_f32=o32, f32_geom=(M * N, N, 0))
            elif kind in ("gated_dual", "gated_f32"):
                kw = dict(gate_tab=keep["tab"], gate=keep["gate"], gate_ld=N, rows_per_gate=100, res=keep["res"],
                          res_geom=(M * N, N, 0), out_f32=o32, f32_geom=(M * N, N, 0))
                if kind == "gated_dual":
                    kw.update(out_act=o16, act_geom=(M * N, N, 0))
            elif kind == "swiglu":
                kw = dict(swiglu=1, out_act=o16, act_geom=(M * n_out, n_out, 0))
            elif kind == "bias_res_batched":
                kw = dict(bias=keep["bias"], res=keep["res"], res_geom=(M * N, N, 0), out_f32=o32, f32_geom=(M * N, N, 0),
                          out_act=o16, act_geom=(M * N, N, 0), nbatch=nb, a_bstride=M * K)
            else:
                kw = dict(alpha=0.37, res=keep["res"], res_geom=(M * N, N, 0), out_f32=o32, f32_geom=(M * N, N, 0))
            util.gemm("bf16", keep["A"], keep["W"], M, N, K, **kw)
            outs[flag] = (o32.cpu(), o16.cpu())
    finally:
        hip.lib().samaudio_debug_set_flag(24, 0)
        hip.lib().samaudio_debug_force_gemm_variant(-1)
    uses32, uses16 = "out_f32" in kw, "out_act" in kw
    for flag in (0, 2, 3):
        if uses32:
            assert torch.isfinite(outs[flag][0]).all()
            assert torch.equal(outs[flag][0], outs[1][0]), flag
        if uses16:
            assert torch.equal(outs[flag][1].view(torch.int16), outs[1][1].view(torch.int16)), flag
            assert outs[flag][1].float().abs().sum() > 0
    # and against the fp32 reference of the simplest forms
    if kind == "f32":
        want = util.rounded(A[0], "bf16") @ util.rounded(W, "bf16").T
        util.report(f"linear epilogue v{variant} f32", outs[0][0][0], want, 5e-4)


@pytest.mark.parametrize("kind", ["act", "gated"])
def test_persistent_tile_walk_is_bitwise_invisible(gpu, kind):
    """gemm8_kernel is a persistent kernel above 256 tiles (at most 256 workgroups, each walking its XCD's run of tiles): a
    launch of 306 tiles - 50 workgroups compute two tiles, the 16-bit epilogue's stores draining under the next tile's
    prologue, the LDS-staged epilogue followed by a barrier - against the one-tile-per-workgroup launch (debug flag 26 = 1):
    identical bits."""
    M, N, K = 4400, 4352, 192
    A, W = _mk((M, K), 71), _mk((N, K), 72, 1 / math.sqrt(K))
    keep = dict(A=util.as_act(A, "bf16", gpu), W=util.as_act(W, "bf16", gpu), tab=_mk((N,), 73).to(gpu),
                gate=_mk((1, N), 74).to(gpu), res=_mk((M, N), 75).to(gpu))
    outs = {}
    try:
        for flag in (0, 1):
            hip.lib().samaudio_debug_force_gemm_variant(22)
            hip.lib().samaudio_debug_set_flag(26, flag)
            o32 = torch.full((M, N), float("nan"), device=gpu)
            o16 = torch.zeros(M, N, device=gpu, dtype=torch.bfloat16)
            kw = dict(out_act=o16, act_geom=(0, N, 0))
            if kind == "gated":
                kw.update(gate_tab=keep["tab"], gate=keep["gate"], gate_ld=N, rows_per_gate=M, res=keep["res"], res_geom=(0, N, 0),
                          out_f32=o32, f32_geom=(0, N, 0))
            util.gemm("bf16", keep["A"], keep["W"], M, N, K, **kw)
            outs[flag] = (o32.cpu(), o16.cpu())
    finally:
        hip.lib().samaudio_debug_set_flag(26, 0)
        hip.lib().samaudio_debug_force_gemm_variant(-1)
    assert torch.equal(outs[0][1].view(torch.int16), outs[1][1].view(torch.int16))
    if kind == "gated":
        assert torch.equal(outs[0][0], outs[1][0]) and torch.isfinite(outs[0][0]).all()
    want = util.rounded(A, "bf16") @ util.rounded(W, "bf16").T
    if kind == "act":
        util.report("persistent walk, 16-bit output", outs[1][1], want, 3.2e-2)
