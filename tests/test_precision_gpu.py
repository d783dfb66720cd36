"""Precision of single GEMM classes (include/samaudio.h SAMAUDIO_OPT_F32_CLASSES / SAMAUDIO_OPT_QUANT_CLASSES).

The reference computes everything in fp32 (README.md:48; no autocast anywhere: SURVEY.md Q17).  The 16-bit engines keep
the GEMM classes whose fp32 cost is negligible - one-row time / modulation GEMMs (transformer.py:490-493), the input and
output projections that touch the ODE state (model.py:116-125, transformer.py:519), the hoisted conditioning and the
y-embedder - on exact-fp32 operands; the fp32 engine can round the operands of chosen classes to a 16-bit format, which
is how tools/error_budget.py attributes the error of a 16-bit mode to classes.
"""
import ctypes as C

import pytest
import torch

from oracle import samaudio_oracle as O
from sam_audio_amd import SAMAudio, hip, preset_config
from sam_audio_amd.synthetic import init_state_dict, synthetic_noise

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def case(gpu):
    cfg = preset_config("mini")
    sd = init_state_dict(cfg, seed=13, with_codec=False)
    B, T, Lt = 2, 60, 5
    g = torch.Generator().manual_seed(3)
    z = torch.randn(B, T, 128, generator=g)
    feats, text = torch.cat([z, z], 2), torch.randn(B, Lt, 768, generator=g)
    tmask = torch.ones(B, Lt, dtype=torch.bool)
    tmask[1, 3:] = False
    pad = torch.ones(B, T, dtype=torch.bool)
    ids, align = O.anchors_to_ids([[("+", 0.4, 1.2)], []], pad, cfg.audio_codec.hop_length, cfg.audio_codec.sample_rate)
    video = torch.randn(B, cfg.vision_encoder.dim, T, generator=g)
    noisy = synthetic_noise(B, T)
    time = torch.tensor([0.3125, 0.3125])

    def field(t, y):
        return O.samaudio_forward(sd, cfg, y, feats, text, t.expand(B), video=video, text_mask=tmask, anchor_ids=ids,
                                  anchor_alignment=align, pad_mask=pad)

    with torch.inference_mode():
        want = O.ode_fixed_grid(field, noisy, method="midpoint", step_size=0.25)
    return dict(cfg=cfg, sd=sd, cond=(feats, text, tmask, video, ids, align, pad), noisy=noisy, want=want)


def _solve(case, gpu, prec, **kw):
    m = SAMAudio(case["cfg"], precision=prec, device=str(gpu), **kw)
    m.load_state_dict(case["sd"], strict=False)
    m._prepare(*case["cond"])
    return m, m.solve(case["noisy"].to(gpu), {"method": "midpoint", "options": {"step_size": 0.25}}).cpu()


def test_f32_classes_run_and_do_not_hurt(gpu, case):
    """bf16 engine, 4 midpoint steps: every fp32-capable class alone and all together.  Each must change the result (the
    fp32 path is really taken), and all together must not be further from the fp32 oracle than none."""
    _, base = _solve(case, gpu, "bf16", f32_classes=0)
    e_none = (base - case["want"]).abs().max().item()
    for cls in ("time", "out", "in", "prep", "yemb"):
        _, lat = _solve(case, gpu, "bf16", f32_classes=cls)
        assert torch.isfinite(lat).all()
        assert not torch.equal(lat, base), f"f32 class '{cls}' left the result bit-identical: path not taken"
        print(f"bf16, f32 class {cls:5s}: max-abs err {(lat - case['want']).abs().max().item():.3e} (none: {e_none:.3e})")
    m, lat = _solve(case, gpu, "bf16")   # default = auto = out + in + prep
    assert m.f32_classes == hip.CLS_F32_DEFAULT
    e_auto = (lat - case["want"]).abs().max().item()
    print(f"bf16 4-step latent: f32_classes none {e_none:.3e} -> auto {e_auto:.3e}")
    assert e_auto <= 1.1 * e_none and e_auto < 2e-2


def test_f32_classes_shard_invariance(gpu, case):
    """rows stay independent with the fp32 classes on: a one-row solve equals that row of the two-row solve bit for bit"""
    m, both = _solve(case, gpu, "bf16")
    one = [None if c is None else c[1:2] for c in case["cond"]]
    m._prepare(*one)
    lat = m.solve(case["noisy"][1:2].to(gpu), {"method": "midpoint", "options": {"step_size": 0.25}}).cpu()
    assert torch.equal(lat[0], both[1])


def test_quantised_classes_emulate_a_16bit_mode(gpu, case):
    m, exact = _solve(case, gpu, "fp32")
    assert (exact - case["want"]).abs().max().item() < 1e-4
    m.set_quantised_classes([], "bf16")
    m._prepare(*case["cond"])
    opt = {"method": "midpoint", "options": {"step_size": 0.25}}
    assert torch.equal(m.solve(case["noisy"].to(gpu), opt).cpu(), exact), "no class selected must be the plain fp32 path"
    errs = {}
    for fmt in ("bf16", "fp16"):
        m.set_quantised_classes(["all"], fmt)
        emu = m.solve(case["noisy"].to(gpu), opt).cpu()
        _, real = _solve(case, gpu, fmt, f32_classes=0)
        e_emu, e_real = (emu - exact).abs().max().item(), (real - exact).abs().max().item()
        errs[fmt] = e_emu
        print(f"{fmt}: all classes rounded in the fp32 engine {e_emu:.3e} vs the real {fmt} engine {e_real:.3e}")
        # same operand roundings of every GEMM; the real engine adds the attention kernels' own 16-bit operands
        assert 0.25 * e_real <= e_emu <= 2.0 * e_real
    assert errs["fp16"] < 0.35 * errs["bf16"]      # 3 more mantissa bits
    m.set_quantised_classes(["w2"], "bf16")
    one = m.solve(case["noisy"].to(gpu), opt).cpu()
    assert 0 < (one - exact).abs().max().item() < errs["bf16"] * 1.2


def test_option_validation(gpu, case):
    m32, _ = _solve(case, gpu, "fp32")
    with pytest.raises(AssertionError):
        hip.check(m32._lib.samaudio_set_option(m32._ctx, hip.OPT_F32_CLASSES, hip.CLS["time"]))
    m16, _ = _solve(case, gpu, "bf16")
    with pytest.raises(AssertionError):
        hip.check(m16._lib.samaudio_set_option(m16._ctx, hip.OPT_F32_CLASSES, hip.CLS["w13"]))   # not fp32-capable
    with pytest.raises(AssertionError):
        hip.check(m16._lib.samaudio_set_option(m16._ctx, hip.OPT_QUANT_CLASSES, 1))
    with pytest.raises(AssertionError):
        hip.check(m32._lib.samaudio_set_option(m32._ctx, hip.OPT_QUANT_FORMAT, 3))
    # a class switched on without its fp32 weight copy fails loudly when the weight set is finalized (naming the copy), not
    # silently in 16 bits and not only at the class's first launch
    m = SAMAudio(case["cfg"], precision="bf16", device=str(gpu), f32_classes="out")
    from sam_audio_amd.weights import convert_dit
    t = convert_dit(case["sd"], case["cfg"], torch.bfloat16, gpu, f32_classes=0)
    assert not any(k.endswith(".f32") for k in t)
    m._register(t)
    with pytest.raises(RuntimeError, match="w_out.f32"):
        hip.check(m._lib.samaudio_finalize(m._ctx, 0))
    # ... and on a finalized context the option itself is refused
    m16b = SAMAudio(case["cfg"], precision="bf16", device=str(gpu), f32_classes=0)
    m16b._register(t)
    hip.check(m16b._lib.samaudio_finalize(m16b._ctx, 0))
    with pytest.raises(RuntimeError, match="t_w13.f32"):
        hip.check(m16b._lib.samaudio_set_option(m16b._ctx, hip.OPT_F32_CLASSES, hip.CLS["time"]))


def test_f32_copies_follow_the_classes_in_use(gpu, case):
    """Only the classes that run in fp32 carry an fp32 operand copy; set_f32_classes adds a class's copies the first time it
    is switched on, and the result equals that of a model built with the class from the start, bit for bit."""
    opt = {"method": "midpoint", "options": {"step_size": 0.25}}
    m = SAMAudio(case["cfg"], precision="bf16", device=str(gpu), f32_classes="out")
    m.load_state_dict(case["sd"], strict=False)
    have = {k for k in m.engine_tensors() if k.endswith(".f32")}
    assert have == {"w_out.f32"}, have
    m.set_f32_classes("out,time,yemb")
    have = {k for k in m.engine_tensors() if k.endswith(".f32")}
    assert have == {"w_out.f32", "t_w13.f32", "t_w2.f32", "tb_w.f32", "y_w13.f32", "y_w2.f32"}, have
    m._prepare(*case["cond"])
    got = m.solve(case["noisy"].to(gpu), opt)
    _, want = _solve(case, gpu, "bf16", f32_classes="out,time,yemb")
    assert torch.equal(got.cpu(), want)
    assert SAMAudio(case["cfg"], precision="bf16", device=str(gpu), f32_classes=None).f32_classes == 0
