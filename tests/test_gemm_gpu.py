"""Parity of the generalised GEMM / implicit-conv kernel (sam_audio_amd/csrc/gemm.hip) through the C ABI.

Checker: plain PyTorch fp32 on the CPU on the same (dtype-rounded) inputs - this is a floating point
kernel, so the tolerance is stated per test: fp32 mode is an exact-fp32 fma chain (<= 2e-4 at K<=4096 on
O(1) data, dominated by summation order), bf16 mode has exact products and fp32 accumulation, so against
bf16-rounded inputs it meets the same bound; outputs stored as bf16 add half an ulp (2^-9 relative).
"""
import math

import pytest
import torch
import torch.nn.functional as F

from sam_audio_amd import hip
from tests import util

pytestmark = pytest.mark.gpu
PRECS = ["fp32", "bf16"]


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K", [(200, 192, 256), (300, 96, 192), (129, 32, 64), (515, 64, 448),
                                   (1000, 384, 1024), (64, 1, 704), (250, 2816, 256)])
def test_plain_gemm_tails_and_tiles(gpu, prec, M, N, K):
    A, W = _mk((M, K), 1), _mk((N, K), 2, 1 / math.sqrt(K))
    out = torch.full((M, N), float("nan"), device=gpu)
    util.gemm(prec, util.as_act(A, prec, gpu), util.as_act(W, prec, gpu), M, N, K, out_f32=out, f32_geom=(0, N, 0))
    want = util.rounded(A, prec) @ util.rounded(W, prec).T
    util.report(f"gemm {prec} {M}x{N}x{K}", out, want, 2e-4)


@pytest.mark.parametrize("prec", PRECS)
def test_asymmetric_identity(gpu, prec):
    """A = I with an asymmetric W catches transposed / permuted output fragments exactly."""
    n = 256
    A = torch.eye(n)
    W = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251) / 8.0  # exactly representable in bf16? no: use /8 of ints<251
    W = W.to(torch.bfloat16).float()
    out = torch.empty(n, n, device=gpu)
    util.gemm(prec, util.as_act(A, prec, gpu), util.as_act(W, prec, gpu), n, n, n, out_f32=out, f32_geom=(0, n, 0))
    assert torch.equal(out.cpu(), W.T.contiguous())


@pytest.mark.parametrize("prec", PRECS)
def test_epilogue_bias_gate_residual_dual_output(gpu, prec):
    B, T, N, K = 3, 70, 256, 320
    M = B * T
    A, W = _mk((M, K), 3), _mk((N, K), 4, 1 / math.sqrt(K))
    bias, tab, gate, res = _mk((N,), 5), _mk((N,), 6), _mk((B, 2 * N), 7), _mk((M, N), 8)
    out = torch.empty(M, N, device=gpu)
    out_act = torch.empty(M, N, device=gpu, dtype=util.ACT_DT[prec])
    gate_d = gate.to(gpu)
    util.gemm(prec, util.as_act(A, prec, gpu), util.as_act(W, prec, gpu), M, N, K, bias=bias.to(gpu),
              gate_tab=tab.to(gpu), gate=gate_d[:, N:].contiguous(), gate_ld=N, rows_per_gate=T, alpha=0.5,
              res=res.to(gpu), res_geom=(0, N, 0), out_f32=out, f32_geom=(0, N, 0), out_act=out_act,
              act_geom=(0, N, 0), act=hip.ACT_SILU)
    acc = util.rounded(A, prec) @ util.rounded(W, prec).T + bias
    g = (tab[None] + gate[:, N:]).repeat_interleave(T, dim=0)
    want = res + 0.5 * g * acc
    util.report(f"epilogue f32 {prec}", out, want, 3e-4)
    util.report(f"epilogue act {prec}", out_act, F.silu(want), 3e-4 if prec == "fp32" else 2e-2)


@pytest.mark.parametrize("prec", PRECS)
def test_epilogue_inplace_residual_shared_gate(gpu, prec):
    """h = h + gate * acc in place with one gate row shared by all rows (n_time == 1)."""
    M, N, K = 333, 128, 128
    A, W = _mk((M, K), 9), _mk((N, K), 10, 1 / math.sqrt(K))
    tab, gate, h = _mk((N,), 11), _mk((N,), 12), _mk((M, N), 13)
    h_d = h.to(gpu)
    util.gemm(prec, util.as_act(A, prec, gpu), util.as_act(W, prec, gpu), M, N, K, gate_tab=tab.to(gpu),
              gate=gate.to(gpu), gate_ld=0, rows_per_gate=50, res=h_d, res_geom=(0, N, 0), out_f32=h_d,
              f32_geom=(0, N, 0))
    want = h + (tab + gate)[None] * (util.rounded(A, prec) @ util.rounded(W, prec).T)
    util.report(f"inplace {prec}", h_d, want, 3e-4)


@pytest.mark.parametrize("prec", PRECS)
def test_swiglu_epilogue(gpu, prec):
    from sam_audio_amd.weights import _interleave16
    M, Fh, K = 270, 192, 256
    A, W1, W3 = _mk((M, K), 14), _mk((Fh, K), 15, 1 / math.sqrt(K)), _mk((Fh, K), 16, 1 / math.sqrt(K))
    W13 = _interleave16(W1, W3)
    out = torch.empty(M, Fh, device=gpu, dtype=util.ACT_DT[prec])
    util.gemm(prec, util.as_act(A, prec, gpu), util.as_act(W13, prec, gpu), M, 2 * Fh, K, swiglu=1, out_act=out,
              act_geom=(0, Fh, 0))
    a = util.rounded(A, prec)
    want = F.silu(a @ util.rounded(W1, prec).T) * (a @ util.rounded(W3, prec).T)
    util.report(f"swiglu {prec}", out, want, 3e-4 if prec == "fp32" else 3e-2)


HALO = 40


def _halo(x_bct, prec, gpu):
    """[B, C, T] -> channels-last halo-padded [B, HALO+T+HALO, C] activation buffer."""
    B, Cc, T = x_bct.shape
    buf = torch.zeros(B, T + 2 * HALO, Cc)
    buf[:, HALO:HALO + T] = x_bct.transpose(1, 2)
    return util.as_act(buf, prec, gpu)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("C_,dil", [(64, 1), (96, 3), (128, 9)])
def test_dilated_conv7_with_snake_residual(gpu, prec, C_, dil):
    """One DAC residual unit: snake -> conv k7 (dilated) -> snake -> conv k1 -> + x, as two launches."""
    from sam_audio_amd.weights import _pad_k
    B, T = 2, 333
    slab = 64 if prec == "bf16" else 32
    x = _mk((B, C_, T), 20)
    w1, b1 = _mk((C_, C_, 7), 21, 1 / math.sqrt(7 * C_)), _mk((C_,), 22, 0.1)
    w2, b2 = _mk((C_, C_, 1), 23, 1 / math.sqrt(C_)), _mk((C_,), 24, 0.1)
    a1, a2, a3 = _mk((C_,), 25, 0.2) + 1, _mk((C_,), 26, 0.2) + 1, _mk((C_,), 27, 0.2) + 1

    def snake(v, a):
        a = a.view(1, -1, 1)
        return v + torch.sin(a * v) ** 2 / (a + 1e-9)

    xs = util.rounded(snake(x, a1), prec)  # what the previous epilogue would have stored
    h = F.conv1d(xs, util.rounded(w1, prec), b1, dilation=dil, padding=3 * dil)
    hs = util.rounded(snake(h, a2), prec)
    y = x + F.conv1d(hs, util.rounded(w2, prec), b2)
    ys = snake(y, a3)

    g1 = util.as_act(_pad_k(w1.permute(0, 2, 1).reshape(C_, 7 * C_), slab), prec, gpu)
    g2 = util.as_act(_pad_k(w2.permute(0, 2, 1).reshape(C_, C_), slab), prec, gpu)
    xin = _halo(snake(x, a1), prec, gpu)
    tmp = torch.zeros_like(xin)
    raw = torch.zeros(B, T + 2 * HALO, C_, device=gpu)
    raw[:, HALO:HALO + T] = x.transpose(1, 2).to(gpu)
    act = torch.zeros_like(xin)
    bs = (T + 2 * HALO) * C_
    geom = (bs, C_, HALO * C_)
    util.gemm(prec, xin, g1, T, C_, g1.shape[1], nbatch=B, a_off=(HALO - 3 * dil) * C_, a_bstride=bs, lda=C_, kc=C_,
              tap_stride=dil * C_, bias=b1.to(gpu), out_act=tmp, act_geom=geom, act=hip.ACT_SNAKE, act_alpha=a2.to(gpu))
    util.gemm(prec, tmp, g2, T, C_, g2.shape[1], nbatch=B, a_off=HALO * C_, a_bstride=bs, lda=C_, kc=C_,
              tap_stride=C_, bias=b2.to(gpu), res=raw, res_geom=geom, out_f32=raw, f32_geom=geom, out_act=act,
              act_geom=geom, act=hip.ACT_SNAKE, act_alpha=a3.to(gpu))
    tol = 5e-4 if prec == "fp32" else 3e-2
    util.report(f"resunit raw {prec} C{C_} d{dil}", raw[:, HALO:HALO + T].transpose(1, 2), y, tol)
    util.report(f"resunit act {prec} C{C_} d{dil}", act[:, HALO:HALO + T].transpose(1, 2), ys, tol)
    assert float(act[:, :HALO].abs().max()) == 0 and float(act[:, HALO + T:].abs().max()) == 0, "halo was written"


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("C_,s", [(64, 2), (128, 8), (32, 12)])
def test_strided_conv(gpu, prec, C_, s):
    B, T = 2, 24 * s
    x = _mk((B, C_, T), 30)
    w, b = _mk((2 * C_, C_, 2 * s), 31, 1 / math.sqrt(2 * s * C_)), _mk((2 * C_,), 32, 0.1)
    want = F.conv1d(util.rounded(x, prec), util.rounded(w, prec), b, stride=s, padding=math.ceil(s / 2))
    To = T // s
    assert want.shape[-1] == To
    g = util.as_act(w.permute(0, 2, 1).reshape(2 * C_, 2 * s * C_), prec, gpu)
    xin = _halo(x, prec, gpu)
    out = torch.zeros(B, To + 2 * HALO, 2 * C_, device=gpu)
    util.gemm(prec, xin, g, To, 2 * C_, 2 * s * C_, nbatch=B, a_off=(HALO - s // 2) * C_, a_bstride=(T + 2 * HALO) * C_,
              lda=s * C_, kc=2 * s * C_, bias=b.to(gpu), out_f32=out,
              f32_geom=((To + 2 * HALO) * 2 * C_, 2 * C_, HALO * 2 * C_))
    util.report(f"strided conv {prec} C{C_} s{s}", out[:, HALO:HALO + To].transpose(1, 2), want, 5e-4)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("Cin,s", [(128, 2), (64, 8), (64, 12)])
def test_transposed_conv(gpu, prec, Cin, s):
    B, T = 2, 37
    Cout = Cin // 2
    x = _mk((B, Cin, T), 40)
    w, b = _mk((Cin, Cout, 2 * s), 41, 1 / math.sqrt(2 * Cin)), _mk((Cout,), 42, 0.1)
    pad = math.ceil(s / 2)
    want = F.conv_transpose1d(util.rounded(x, prec), util.rounded(w, prec), b, stride=s, padding=pad)
    To = T * s
    assert want.shape[-1] == To
    wg = torch.stack([w[:, :, s:], w[:, :, :s]], dim=0).permute(3, 2, 0, 1).reshape(s * Cout, 2 * Cin)
    g = util.as_act(wg, prec, gpu)
    xin = _halo(x, prec, gpu)
    out = torch.zeros(B, To + 2 * HALO, Cout, device=gpu)
    util.gemm(prec, xin, g, T + 1, s * Cout, 2 * Cin, nbatch=B, a_off=(HALO - 1) * Cin,
              a_bstride=(T + 2 * HALO) * Cin, lda=Cin, kc=2 * Cin, bias=b.to(gpu), chan_mod=Cout, out_f32=out,
              f32_geom=((To + 2 * HALO) * Cout, s * Cout, (HALO - pad) * Cout), c_lo=pad * Cout,
              c_hi=(To + pad) * Cout, c_ld_rel=s * Cout)
    util.report(f"convT {prec} Cin{Cin} s{s}", out[:, HALO:HALO + To].transpose(1, 2), want, 5e-4)
    assert float(out[:, :HALO].abs().max()) == 0 and float(out[:, HALO + To:].abs().max()) == 0, "halo was written"


@pytest.mark.parametrize("prec", PRECS)
def test_patcher_conv3(gpu, prec):
    B, T, D = 2, 50, 256
    x = _mk((B, D, T), 50)
    w, b = _mk((D, D, 3), 51, 1 / math.sqrt(3 * D)), _mk((D,), 52, 0.1)
    want = F.conv1d(util.rounded(x, prec), util.rounded(w, prec), b, padding=1)
    buf = torch.zeros(B, T + 2, D)
    buf[:, 1:T + 1] = x.transpose(1, 2)
    g = util.as_act(w.permute(0, 2, 1).reshape(D, 3 * D), prec, gpu)
    out = torch.empty(B, T, D, device=gpu)
    util.gemm(prec, util.as_act(buf, prec, gpu), g, T, D, 3 * D, nbatch=B, a_off=0, a_bstride=(T + 2) * D, lda=D, kc=D,
              tap_stride=D, bias=b.to(gpu), out_f32=out, f32_geom=(T * D, D, 0))
    util.report(f"patcher conv {prec}", out.transpose(1, 2), want, 5e-4)


def test_gemm_rejects_bad_k(gpu):
    A = torch.zeros(64, 48, device=gpu)
    with pytest.raises(AssertionError):
        util.gemm("fp32", A, A, 64, 64, 48, out_f32=A, f32_geom=(0, 64, 0))
