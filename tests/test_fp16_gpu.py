"""The fp16-operand build of the library (libsamaudio_hip_f16.so = the same sources with -DSA_OPERAND_FP16; host classes
select it with precision="fp16").  Same kernels, same MFMA rate on gfx950, 10 instead of 7 mantissa bits per operand
rounding: VERDICT round 1, item 1(d) - a 16-bit mode that gets close to the north_star's 1e-3.

Checks: (i) the GEMM kernels themselves on fp16-rounded operands (every shipped tile family through the C ABI of the
fp16 library); (ii) whole separate() at 'mini' dims: latent and waveform error against the fp32 CPU oracle, next to the
bf16 figure.  The large*-dims forward / 2-step solve in fp16 are in tests/test_large_gpu.py.
"""
import math

import pytest
import torch

from oracle import samaudio_oracle as O
from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config
from sam_audio_amd.synthetic import init_state_dict, synthetic_clip, synthetic_noise, synthetic_text_features
from tests import util

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif("SAMAUDIO_EMU_DRYRUN" in __import__("os").environ,
                                 reason="the CPU dry-run builds carry the bf16 operand format only")]


def _mk(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(autouse=True)
def _restore_variant():
    yield
    hip.lib("fp16").samaudio_debug_force_gemm_variant(-1)
    hip.lib("fp16").samaudio_debug_set_flag(27, 0)


@pytest.mark.parametrize("variant", [-1, 1, 22, 25, 26, 27])
def test_gemm_kernels_on_fp16_operands(gpu, variant):
    """fp16 x fp16 products are exact in fp32 and the accumulation is fp32: fp32 outputs agree with a CPU fp32 matmul
    on the same fp16-rounded operands to summation-order noise; fp16 outputs add half an fp16 ulp (2^-11 relative)."""
    hip.lib("fp16").samaudio_debug_force_gemm_variant(variant)
    B, T, N, K = 3, 90, 384, 448
    M = B * T
    A, W = _mk((M, K), 21), _mk((N, K), 22, 1 / math.sqrt(K))
    tab, gate, res, bias = _mk((N,), 23), _mk((B, N), 24), _mk((M, N), 25), _mk((N,), 26)
    keep = [tab.to(gpu), gate.to(gpu), res.to(gpu), bias.to(gpu)]
    out = torch.full((M, N), float("nan"), device=gpu)
    out_act = torch.zeros(M, N, device=gpu, dtype=torch.float16)
    util.gemm("fp16", util.as_act(A, "fp16", gpu), util.as_act(W, "fp16", gpu), M, N, K, bias=keep[3], gate_tab=keep[0],
              gate=keep[1], gate_ld=N, rows_per_gate=T, res=keep[2], res_geom=(0, N, 0), out_f32=out, f32_geom=(0, N, 0),
              out_act=out_act, act_geom=(0, N, 0), act=hip.ACT_SILU)
    base = util.rounded(A, "fp16") @ util.rounded(W, "fp16").T + bias
    want = base * (tab[None] + gate.repeat_interleave(T, 0)) + res
    util.report(f"fp16 gemm v{variant} f32", out, want, 5e-4)
    util.report(f"fp16 gemm v{variant} act", out_act, torch.nn.functional.silu(want), 8.5e-3)  # half an fp16 ulp at |v| in [16, 32)


def test_separate_fp16_error_next_to_bf16(gpu):
    cfg = preset_config("mini")
    sd = init_state_dict(cfg, seed=8)
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 25 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 8)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["x", "y"], audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(2, 25)
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise)
    errs = {}
    for prec in ("bf16", "fp16"):
        model = SAMAudio(cfg, precision=prec, device=str(gpu))
        model.load_state_dict(sd, strict=False)
        res = model.separate(batch.to(gpu), noise=noise.to(gpu))
        lat = (model.last_latent.cpu() - lat_ref).abs().max().item()
        wav = max((a.cpu() - b).abs().max().item() for a, b in zip(res.target + res.residual, t_ref + r_ref))
        errs[prec] = (lat, wav)
    print(f"full 16-step midpoint ODE + decode, 'mini' dims (latent max {lat_ref.abs().max().item():.2f}): "
          f"bf16 latent {errs['bf16'][0]:.3e} waveform {errs['bf16'][1]:.3e} | "
          f"fp16 latent {errs['fp16'][0]:.3e} waveform {errs['fp16'][1]:.3e}")
    # measured on MI355X (profiles/r2_call7/gpu_tests.log): fp16 latent 7.1e-4, waveform 1.6e-4 (bf16: 6.1e-3 / 1.5e-3);
    # bounds = 2 x measured, both inside the north_star's 1e-3 ... and 8x below bf16
    assert errs["fp16"][0] < 1.5e-3 and errs["fp16"][1] < 3.5e-4
    assert errs["fp16"][0] < errs["bf16"][0]


@pytest.mark.parametrize("variant,roles", [(22, 0), (27, 0), (27, 1), (27, 2), (27, 3)])
@pytest.mark.parametrize("kind", ["act", "swiglu", "gated"])
def test_mixed_mode_gemm_reads_and_writes_bfloat16_inside_the_fp16_library(gpu, variant, roles, kind):
    """precision="mixed" (samaudio.h SAMAUDIO_OPT_ALT16_CLASSES): the five big GEMM classes run on bfloat16 operands inside
    the fp16 build.  GemmParams.flags bit 10: A and W are bfloat16 (the bf16 MFMA); bit 9: the 16-bit output is written as
    bfloat16 (it feeds another such GEMM).  Against fp32 torch on the bf16-rounded operands; outputs decoded per flag."""
    import ctypes as C
    lib = hip.lib("fp16")
    lib.samaudio_debug_force_gemm_variant(variant)
    lib.samaudio_debug_set_flag(27, roles)   # gemm8s' pipelined form: without / with requesting waves (gemm8.hip)
    M, N, K = 333, 512, 320
    A, W = _mk((M, K), 31), _mk((N, K), 32, 1 / math.sqrt(K))
    Ab, Wb = A.to(torch.bfloat16).to(gpu), W.to(torch.bfloat16).to(gpu)
    prod = A.to(torch.bfloat16).float() @ W.to(torch.bfloat16).float().T
    for out_alt in (0, 1):
        n_out = N // 2 if kind == "swiglu" else N
        o16 = torch.zeros(M, n_out, device=gpu, dtype=torch.bfloat16 if out_alt else torch.float16)
        o32 = torch.full((M, N), float("nan"), device=gpu)
        kw = dict(out_act=o16, act_geom=(0, n_out, 0))
        want = prod
        if kind == "swiglu":
            kw["swiglu"] = 1
            w1 = prod.view(M, N // 32, 2, 16)[:, :, 0].reshape(M, N // 2)   # 16-row interleave of w1 / w3 (weights.py)
            w3 = prod.view(M, N // 32, 2, 16)[:, :, 1].reshape(M, N // 2)
            want = torch.nn.functional.silu(w1) * w3
        elif kind == "gated":
            tab, gate, res = _mk((N,), 33).to(gpu), _mk((1, N), 34).to(gpu), _mk((M, N), 35).to(gpu)
            kw.update(gate_tab=tab, gate=gate, gate_ld=N, rows_per_gate=M, res=res, res_geom=(0, N, 0), out_f32=o32,
                      f32_geom=(0, N, 0))
            want = prod * (tab.cpu()[None] + gate.cpu()) + res.cpu()
        p = util.gemm_params(Ab, Wb, M, N, K, **kw)
        p.flags = 1024 | (512 if out_alt else 0)
        hip.check(lib.samaudio_op_gemm(C.byref(p), C.sizeof(p), util.PREC["fp16"], util.stream()))
        tol16 = want.abs().max().item() * (2.0 ** -8 if out_alt else 2.0 ** -11) * 1.02   # half an ulp of the output format
        util.report(f"mixed gemm v{variant} {kind} out_alt={out_alt} 16-bit", o16, want, tol16)
        if kind == "gated":
            util.report(f"mixed gemm v{variant} {kind} out_alt={out_alt} f32", o32, want, 5e-4)


def test_separate_mixed_precision_error_next_to_fp16(gpu):
    """precision="mixed" end to end at 'mini' dims: bf16 operands on the five big GEMM classes of the DiT layers, fp16 elsewhere -
    latent and waveform error against the fp32 CPU oracle, between the fp16 and the bf16 figures."""
    cfg = preset_config("mini")
    sd = init_state_dict(cfg, seed=8)
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 25 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 5, ragged=True)
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["x"] * 2, audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(2, 25)
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise)
    errs = {}
    for prec in ("fp16", "mixed", "bf16"):
        m = SAMAudio(cfg, precision=prec, device=str(gpu))
        m.load_state_dict(sd, strict=False)
        res = m.separate(batch.to(gpu), noise=noise.to(gpu))
        errs[prec] = ((m.last_latent.cpu() - lat_ref).abs().max().item(),
                      max((a.cpu() - b).abs().max().item() for a, b in zip(res.target + res.residual, t_ref + r_ref)))
        print(f"mini separate {prec}: latent err {errs[prec][0]:.3e}, waveform err {errs[prec][1]:.3e}")
    assert errs["mixed"][0] < errs["bf16"][0] and errs["mixed"][0] < 5 * errs["fp16"][0] + 1e-3
    assert errs["mixed"][1] < 2 * errs["fp16"][1] + 1e-4
