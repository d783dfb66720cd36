"""Compensated 16-bit GEMM operands: precision "fp16x3" / "bf16x3" (include/samaudio.h SAMAUDIO_OPT_X3_CLASSES).

The reference computes separate() in fp32 (README.md:48).  No plain 16-bit operand format holds the north_star's 1e-3 max-abs on
trained-like weight statistics (tests/test_hostile_gpu.py); the x3 modes keep fp32 storage and run the six big GEMM classes of
the DiT layers (reference transformer.py:121-161,195-206,354-391) on hi/lo-split operands: one 16-bit MFMA GEMM over K' = 3K.
Here: the split kernel, the GEMM identity through the C ABI (incl. halves that are SUBNORMAL in IEEE half: the MFMA must not flush
them), and the full path against the CPU oracle.  On the CPU simulator (SAMAUDIO_EMU_DRYRUN=simt: bfloat16 library only) the
bf16x3 form exercises the same kernels, layouts and engine plumbing.
"""
import ctypes as C
import os

import pytest
import torch

from oracle import samaudio_oracle as O
from sam_audio_amd import SAMAudio, SAMAudioProcessor, hip, preset_config
from sam_audio_amd.synthetic import init_state_dict, make_hostile, synthetic_clip, synthetic_noise, synthetic_text_features
from sam_audio_amd.weights import fly16_to_f32, fly16_weight, ktm_to_rows, x3_weight
from tests import util

pytestmark = pytest.mark.gpu
SIM = os.environ.get("SAMAUDIO_EMU_DRYRUN", "") != ""
X3 = ["bf16x3"] if SIM else ["fp16x3", "bf16x3"]
HALF = {"fp16x3": torch.float16, "bf16x3": torch.bfloat16}
PLAIN = {"fp16x3": "fp16", "bf16x3": "bf16"}   # the plain 16-bit mode of the same library


def _split3(x, prec, gpu):
    M, K = x.shape
    out = torch.empty(M, 3 * K, dtype=HALF[prec], device=gpu)
    xd = x.to(gpu).contiguous()
    hip.check(hip.lib(hip.operands_for(prec)).samaudio_op_split3(hip.ptr(xd), K, hip.ptr(out), M, K, util.stream()))
    return out


@pytest.mark.parametrize("prec", X3)
def test_split3_writes_lo_hi_hi(gpu, prec):
    g = torch.Generator().manual_seed(1)
    M, K = 37, 192
    x = torch.randn(M, K, generator=g) * torch.logspace(-6, 3, K)[None, :]   # 1e-6 .. 1e3: lo halves from subnormal to large
    x[0, :8] = torch.tensor([0.0, -0.0, 65504.0, 7e4, -1.2e5, 6e-8, 1.0, -3.3333333])
    half = HALF[prec]
    out = _split3(x, prec, gpu).cpu()
    hi_ref = (x.clamp(-65504.0, 65504.0) if half == torch.float16 else x).to(half)
    lo_ref = (x - hi_ref.float()).to(half)
    assert torch.equal(out[:, K:2 * K], hi_ref) and torch.equal(out[:, 2 * K:], hi_ref), "hi halves"
    assert torch.equal(out[:, :K], lo_ref), "lo half"
    # what the pair represents: x to ~2^-22 (IEEE half; coarser only where lo itself is subnormal) / 2^-17 (bfloat16)
    fin = x.abs() < (1.3e5 if half == torch.float16 else 1e38)
    err = (out[:, :K].float() + out[:, K:2 * K].float() - x).abs()[fin]
    # relative 2^-21 / 2^-15, or - where lo itself is subnormal in IEEE half - half its quantum 2^-24
    bound = (x.abs() * (2.0 ** -21 if half == torch.float16 else 2.0 ** -15)).clamp_min(2.0 ** -25 if half == torch.float16 else 0.0)[fin]
    print(f"split3 {prec}: max |hi + lo - x| / bound = {(err / bound.clamp_min(1e-45)).max().item():.3f}")
    assert (err <= bound).all()


@pytest.mark.parametrize("prec", X3)
@pytest.mark.parametrize("shape", [(300, 256, 128), (130, 512, 448)])
def test_x3_gemm_is_the_fp32_product(gpu, prec, shape):
    """[x_lo | x_hi | x_hi] . [W_hi | W_lo | W_hi]^T through the library's 16-bit GEMM, both weight layouts, against the fp64 product
    of the fp32 operands - and against what the plain 16-bit operands give.  Column scales put many lo halves into IEEE half's
    SUBNORMAL range (|lo| < 6.1e-5): a matrix core that flushed them would lose the compensation."""
    M, N, K = shape
    g = torch.Generator().manual_seed(2)
    x = torch.randn(M, K, generator=g) * torch.logspace(-2, 1, K)[None, :]
    w = torch.randn(N, K, generator=g) * 0.05
    ref = (x.double() @ w.double().T).float()
    half = HALF[prec]
    plain = (x.to(half).double() @ w.to(half).double().T).float()
    a3 = _split3(x, prec, gpu)
    errs = {}
    for ktm in (False, True):
        w3 = x3_weight(w, half, ktm=ktm).to(gpu)
        if ktm:
            assert torch.equal(ktm_to_rows(w3.cpu()), x3_weight(w, half, ktm=False))
        out = torch.full((M, N), float("nan"), device=gpu)
        util.gemm(PLAIN[prec], a3, w3, M, N, 3 * K, out_f32=out, f32_geom=(0, N, 0), flags=2048 if ktm else 0)
        errs[ktm] = (out.cpu() - ref).abs().max().item()
    e_plain = (plain - ref).abs().max().item()
    print(f"x3 GEMM {prec} {shape}: max-abs err rows {errs[False]:.3e} / ktm {errs[True]:.3e}; plain 16-bit operands {e_plain:.3e}; "
          f"|ref| <= {ref.abs().max():.2f}")
    assert errs[False] == errs[True], "the two weight layouts accumulate in the same order"
    assert errs[True] < e_plain / (200 if half == torch.float16 else 20)
    assert errs[True] < 2e-5 * ref.abs().max().item() if half == torch.float16 else True


@pytest.mark.parametrize("prec", X3)
@pytest.mark.parametrize("shape", [(300, 96, 672, 96), (200, 192, 256, 256), (70, 1, 448, 64), (150, 128, 448, 64), (140, 64, 448, 64)])
def test_fp32_gemm_with_operands_split_on_the_fly(gpu, prec, shape):
    """GemmParams.flags bit 13 (common.h GEMM_FLAG_X3_FLY; SAMAUDIO_OPT_X3_CLASSES bit CODEC): the fp32 kernel of gemm.hip splits the
    fp32 fragments of BOTH operands in registers and multiplies on the 16-bit MFMA - here as a dilated implicit convolution
    (kc < K: the DAC-VAE's k7 convolutions, reference codec.py:65-89) with bias and Snake, against the exact-fp32 launch of the same
    parameters and the fp64 product."""
    M, N, K, kc = shape
    g = torch.Generator().manual_seed(5)
    taps, dil, halo = K // kc, 3, 40
    x = torch.randn(M + 2 * halo, kc, generator=g) * torch.logspace(-1.5, 1, kc)[None, :]
    w = torch.randn(N, K, generator=g) * 0.05
    bias, alpha = torch.randn(N, generator=g) * 0.1, torch.rand(N, generator=g) + 0.5
    a_off = (halo - (taps // 2) * dil) * kc
    rows = torch.stack([x[halo - (taps // 2) * dil + j * dil: halo - (taps // 2) * dil + j * dil + M] for j in range(taps)], 1).reshape(M, K)
    ref = rows.double() @ w.double().T + bias.double()
    lib = hip.lib(hip.operands_for(prec))
    outs = {}
    wfly = fly16_weight(w, HALF[prec]).to(gpu)   # flags bit 14: the weight already split, in the kernel's fragment layout
    assert torch.equal(fly16_to_f32(wfly.cpu()), w.to(HALF[prec]).float() + (w - w.to(HALF[prec]).float()).to(HALF[prec]).float())
    # (flags, debug flag 36): exact fp32 | both operands split in registers | the split weight on the 4 x 1-wave tiles (shipped) | on the
    # tiles of the plain policy
    cases = {"exact": (0, 0), "fly": (8192, 0), "twin": (8192 | 16384, 0), "twin_old_tiles": (8192 | 16384, 1)}
    for name, (flags, dbg) in cases.items():
        o32, oact = torch.full((M, N), float("nan"), device=gpu), torch.full((M, N), float("nan"), device=gpu)
        xd, wd, bd, ad = x.to(gpu).contiguous(), (wfly if flags & 16384 else w.to(gpu).contiguous()), bias.to(gpu), alpha.to(gpu)
        prm = util.gemm_params(xd, wd, M, N, K, a_off=a_off, lda=kc, kc=kc, tap_stride=dil * kc, bias=bd, out_f32=o32, f32_geom=(0, N, 0),
                               out_act=oact, act_geom=(0, N, 0), act=hip.ACT_SNAKE, act_alpha=ad, flags=flags)
        lib.samaudio_debug_set_flag(36, dbg)
        try:
            hip.check(lib.samaudio_op_gemm(C.byref(prm), C.sizeof(prm), hip.F32, util.stream()))
            outs[name] = (o32.cpu(), oact.cpu())
        finally:
            lib.samaudio_debug_set_flag(36, 0)
    for name in cases:
        if name.startswith("twin"):
            assert torch.equal(outs[name][0], outs["fly"][0]) and torch.equal(outs[name][1], outs["fly"][1]), f"{name}: same bits as the register split"
    e_exact = (outs["exact"][0] - ref.float()).abs().max().item()
    e_fly = (outs["fly"][0] - ref.float()).abs().max().item()
    half = HALF[prec]
    e_plain = ((rows.to(half).double() @ w.to(half).double().T + bias.double()).float() - ref.float()).abs().max().item()
    snake = ref + torch.sin(alpha.double() * ref) ** 2 / (alpha.double() + 1e-9)
    e_act = (outs["fly"][1] - snake.float()).abs().max().item()
    print(f"fp32 GEMM, operands split on the fly ({prec}) {shape}: max-abs err {e_fly:.3e} (exact-fp32 launch {e_exact:.3e}, plain 16-bit "
          f"operands {e_plain:.3e}); Snake output {e_act:.3e}; |ref| <= {ref.abs().max():.2f}")
    tol = (4e-6 if half == torch.float16 else 1e-4) * max(1.0, ref.abs().max().item())
    assert e_fly < tol and e_act < 2 * tol and e_fly < e_plain / 20


@pytest.mark.parametrize("prec", X3)
def test_codec_roundtrip_in_x3_context(gpu, prec):
    """DAC-VAE encode and decode of an x3 model (fp32 context, convolutions on operands split on the fly; decoder likewise unless
    codec_decode='16') against the oracle (reference codec.py:65-89)."""
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=3)
    hop = cfg.audio_codec.hop_length
    wav = torch.stack([synthetic_clip(i, 20 * hop) for i in range(2)])
    with torch.inference_mode():
        z_ref = O.dac_encode(sd, cfg.audio_codec, wav)
        w_ref = O.dac_decode(sd, cfg.audio_codec, z_ref).squeeze(1)
    for dec in ("32", "16"):
        model = SAMAudio(cfg, precision=prec, device=str(gpu), codec_decode=dec)
        model.load_state_dict(sd, strict=False)
        z = model.encode_audio(wav.to(gpu))
        w = model.decode_audio(z)
        e_z = (z.cpu() - z_ref.transpose(1, 2)).abs().max().item()
        e_w = (w.cpu() - w_ref).abs().max().item()
        print(f"codec in an {prec} model, decoder '{dec}': encode latent err {e_z:.3e} (|z| <= {z_ref.abs().max():.2f}), decoded waveform err "
              f"{e_w:.3e} (|w| <= {w_ref.abs().max():.2f})")
        assert e_z < (2e-5 if prec == "fp16x3" else 5e-4)
        assert e_w < ((2e-5 if prec == "fp16x3" else 5e-4) if dec == "32" else 5e-3)


@pytest.mark.parametrize("prec", X3)
@pytest.mark.parametrize("T", [50, 250, 300])
def test_self_attention_on_split_operands(gpu, prec, T):
    """SAMAUDIO_X3_ATTENTION: fp32 Q / K / V^T in, fp32 rows out, both contractions on hi/lo-split operands on the 16-bit MFMA -
    against torch fp64 softmax attention (reference transformer.py:153-160), and against what 16-bit operands would give."""
    import math
    B, H = 2, 2
    Tp = (T + 63) // 64 * 64
    D = H * 128
    g = torch.Generator().manual_seed(20 + T)
    q, k, v = (torch.randn(B, H, T, 128, generator=g) for _ in range(3))
    q = q * 1.5
    mask = torch.ones(B, T, dtype=torch.bool)
    mask[1, T - 13:] = False
    pad = lambda z: torch.nn.functional.pad(z, (0, 0, 0, Tp - T))
    qd, kd = pad(q).contiguous().to(gpu), pad(k).contiguous().to(gpu)
    vtd = pad(v).transpose(2, 3).contiguous().to(gpu)
    out = torch.full((B * T, D), float("nan"), device=gpu)
    md = mask.to(gpu).to(torch.uint8)   # (kept alive across the call: the library borrows the pointer)
    hip.check(hip.lib(hip.operands_for(prec)).samaudio_op_self_attention(
        hip.ptr(qd), hip.ptr(kd), hip.ptr(vtd), hip.ptr(md), hip.ptr(out), 2, B, T, Tp, H, util.stream()))

    def ref(qq, kk, vv, round_p=None):
        s = (qq.double() @ kk.double().transpose(-1, -2)) / math.sqrt(128)
        s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
        p = torch.softmax(s, -1)
        if round_p is not None:
            p = p.to(round_p).double()
        return (p @ vv.double()).permute(0, 2, 1, 3).reshape(B * T, D).float()

    want = ref(q, k, v)
    half = HALF[prec]
    plain = ref(q.to(half), k.to(half), v.to(half), half)
    err, e_plain = (out.cpu() - want).abs().max().item(), (plain - want).abs().max().item()
    print(f"self-attention on split operands {prec} T={T}: max-abs err {err:.3e}; plain 16-bit operands {e_plain:.3e}")
    assert err < (2e-5 if half == torch.float16 else 2e-4) and err < e_plain / 10


@pytest.mark.parametrize("prec", X3)
def test_separate_x3_matches_oracle(gpu, prec):
    """Full path at 'mini' dims (ragged text mask, anchors): the x3 mode must sit with the fp32 mode, far inside 1e-3, where the
    plain 16-bit mode of the same library does not."""
    cfg = preset_config("mini")
    sd = init_state_dict(cfg, seed=8)
    hop = cfg.audio_codec.hop_length
    T = 12 if SIM else 25
    clips = [synthetic_clip(i, T * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 8, ragged=True)
    anchors = [[("+", 0.04, 0.12)], [("-", 0.0, 0.08), ("+", 0.08, 0.16)]]
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["x", "y"], audios=clips, anchors=anchors, text_features=text,
                                               text_mask=tmask)
    noise = synthetic_noise(2, T)
    steps = 2 if SIM else 16
    opt = {"method": "midpoint", "options": {"step_size": 1.0 / steps}}
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise, anchors=anchors,
                                           step_size=1.0 / steps)
    errs = {}
    for p in (prec, PLAIN[prec]):
        model = SAMAudio(cfg, precision=p, device=str(gpu))
        model.load_state_dict(sd, strict=False)
        res = model.separate(batch.to(gpu), noise=noise.to(gpu), ode_opt=opt)
        lat = (model.last_latent.cpu() - lat_ref).abs().max().item()
        wav = max((a.cpu() - b).abs().max().item() for a, b in zip(res.target + res.residual, t_ref + r_ref))
        errs[p] = (lat, wav)
    print(f"separate 'mini' {steps} midpoint steps: {prec} latent {errs[prec][0]:.3e} wave {errs[prec][1]:.3e}; "
          f"{PLAIN[prec]} latent {errs[PLAIN[prec]][0]:.3e} wave {errs[PLAIN[prec]][1]:.3e} (|latent| <= {lat_ref.abs().max():.2f})")
    tol = 1e-3 if prec == "fp16x3" else 2e-3   # bfloat16 halves: 16 mantissa bits per operand
    assert errs[prec][0] < tol and errs[prec][1] < tol
    assert errs[prec][0] < errs[PLAIN[prec]][0] / 4


@pytest.mark.parametrize("prec", X3)
@pytest.mark.parametrize("text_len", [3, 12, 20])
def test_x3_cross_attention_folds_for_short_memories_only(gpu, prec, text_len):
    """x3 context, class CWO: text memories of <= 16 tokens take the folded form on compensated operands (h += P . U over K' = 3 * 192:
    cross_attn_probs3 / cross_attn_fold3 kernels, 8- and 16-token head slots), longer ones the unfolded c_wo GEMM over K' = 3 D - both
    against the oracle (reference transformer.py:382-388), with a ragged text mask."""
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=5)
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 6 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, text_len, ragged=True)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["a", "b"], audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(2, 6)
    opt = {"method": "euler", "options": {"step_size": 0.5}}
    with torch.inference_mode():
        _, _, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise, method="euler", step_size=0.5,
                                   decode=False)
    model = SAMAudio(cfg, precision=prec, device=str(gpu))
    model.load_state_dict(sd, strict=False)
    model.separate(batch.to(gpu), noise=noise.to(gpu), ode_opt=opt)
    err = (model.last_latent.cpu() - lat_ref).abs().max().item()
    print(f"x3 cross-attention, {text_len} text tokens ({prec}): latent max-abs err {err:.3e} (|ref| <= {lat_ref.abs().max():.2f})")
    assert err < (1e-4 if prec == "fp16x3" else 1e-3)


@pytest.mark.parametrize("prec", X3)
def test_x3_classes_can_be_switched_per_class_and_need_their_weights(gpu, prec):
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=3)
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 6 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 4)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["a", "b"], audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(2, 6)
    opt = {"method": "euler", "options": {"step_size": 0.5}}
    with torch.inference_mode():
        _, _, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise, method="euler", step_size=0.5,
                                   decode=False)
    lats = {}
    for classes in ("auto", "w13,w2", "qkv,wo,cwq,cwo"):
        model = SAMAudio(cfg, precision=prec, device=str(gpu), x3_classes=classes)
        model.load_state_dict(sd, strict=False)
        model.separate(batch.to(gpu), noise=noise.to(gpu), ode_opt=opt)
        lats[classes] = model.last_latent.cpu()
        assert (lats[classes] - lat_ref).abs().max().item() < 1e-3
    assert not torch.equal(lats["auto"], lats["w13,w2"])   # the masks really select different kernels
    # a class without its split weights is refused with the tensor's name (RuntimeError = SAMAUDIO_ERR_WEIGHT)
    model = SAMAudio(cfg, precision=prec, device=str(gpu), x3_classes="w2")
    model.load_state_dict(sd, strict=False)
    model.x3_classes = hip.CLS_X3_DEFAULT
    with pytest.raises(RuntimeError, match=r"\.x3"):
        model._set_precision_options(model._ctx)


@pytest.mark.skipif(SIM, reason="10 s clips at small* dims: hardware only")
def test_x3_holds_the_bound_on_hostile_weights_where_fp16_does_not(gpu):
    """tests/test_hostile_gpu.py's configuration (trained-like statistics, separate() as timed) - the parity gate of the headline mode."""
    size = os.environ.get("SAMAUDIO_HOSTILE_SIZE", "small*")
    cfg = preset_config(size)
    sd = make_hostile(init_state_dict(cfg, seed=0, device=gpu), cfg, seed=0)
    sd_cpu = {k: v.cpu() for k, v in sd.items()}
    R = 2
    n = 10 * cfg.audio_codec.sample_rate // cfg.audio_codec.hop_length * cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, n) for i in range(R)]
    text, tmask = synthetic_text_features(R, 8, seed=7)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["sound"] * R, audios=clips, text_features=text, text_mask=tmask)
    noise = synthetic_noise(R, n // cfg.audio_codec.hop_length)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    with torch.inference_mode():
        t_ref, r_ref, lat_ref = O.separate(sd_cpu, cfg, batch.audios, batch.sizes.long(), text, tmask, noise)
    model = SAMAudio(cfg, precision="fp16x3", device=str(gpu))
    model.load_state_dict(sd, strict=False)
    res = model.separate(batch.to(gpu), noise=noise.to(gpu))
    lat = (model.last_latent.cpu() - lat_ref).abs().max().item()
    wav = max((a.cpu() - b).abs().max().item() for a, b in zip(res.target + res.residual, t_ref + r_ref))
    print(f"hostile {size} fp16x3: latent max-abs err {lat:.3e} (|ref| <= {lat_ref.abs().max():.2f}), waveform {wav:.3e}")
    assert lat <= 1e-3 and wav <= 1e-3


@pytest.mark.parametrize("prec", X3)
@pytest.mark.parametrize("shape", [(300, 256, 128), (130, 512, 448), (520, 768, 64)])
def test_x3_gemm_sharing_the_operand_tiles(gpu, prec, shape):
    """GemmParams.flags bit 15 (common.h GEMM_FLAG_X3_SHARE): the 8-phase kernels walk the K-concatenated split operands product by
    product per original K-tile - (x_lo, W_hi) (x_hi, W_hi) (x_hi, W_lo) - sharing the operand tiles two consecutive products have in
    common (gemm8x_kernel), the 128 x 128 kernel in the same order through its K-tile index map: the fp32 product as before, and THE
    SAME BITS from both tile sizes (the tile policy picks by row count: batch-sharding invariance), in both weight layouts."""
    M, N, K = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn(M, K, generator=g) * torch.logspace(-2, 1, K)[None, :]
    w = torch.randn(N, K, generator=g) * 0.05
    ref = (x.double() @ w.double().T).float()
    half = HALF[prec]
    a3 = _split3(x, prec, gpu)
    lib = hip.lib(hip.operands_for(prec))
    outs = {}
    try:
        for ktm in (False, True):
            w3 = x3_weight(w, half, ktm=ktm).to(gpu)
            for variant in (22, 27):
                for share in (0, 32768):
                    lib.samaudio_debug_force_gemm_variant(variant)
                    out = torch.full((M, N), float("nan"), device=gpu)
                    util.gemm(PLAIN[prec], a3, w3, M, N, 3 * K, out_f32=out, f32_geom=(0, N, 0), flags=(2048 if ktm else 0) | share)
                    outs[(ktm, variant, share)] = out.cpu()
    finally:
        lib.samaudio_debug_force_gemm_variant(-1)
    plain = outs[(False, 22, 0)]
    shared = outs[(False, 22, 32768)]
    e_plain, e_shared = (plain - ref).abs().max().item(), (shared - ref).abs().max().item()
    print(f"x3 GEMM {prec} {shape}: max-abs err plain walk {e_plain:.3e}, shared operand tiles {e_shared:.3e}; |ref| <= {ref.abs().max():.2f}")
    for key, o in outs.items():
        assert torch.equal(o, shared if key[2] else plain), f"{key}: the order of a walk does not depend on tile size or weight layout"
    assert e_shared <= 2 * e_plain + 1e-6 * ref.abs().max().item()
