"""GPU parity of the reranking / span-prediction rows against oracle/judge_oracle.py
(pinned to the reference's own SAMAudioJudgeModel.forward and to the Hugging Face PeAudioEncoder on the CPU).

STATUS: written in the GPU-less tail of round 1 (the round's 90 GPU-minutes were spent on the separate() path), so
these tests had not run on hardware when they were committed.  What was verified instead, on the CPU: every test of
this file passes in the emulation dry run (`SAMAUDIO_EMU_DRYRUN=1 python -m pytest tests/test_zz_next_rows_gpu.py -m gpu`,
see tests/conftest.py: the product's host classes bound to the unchanged host orchestration sources linked against
emulated kernel launchers), as do 14 of the 15 tests of test_path_gpu.py that are green on MI355X (the 15th needs real
streams).  What only hardware can show are the five new streaming kernels of csrc/peav_kernels.hip and the verified
kernels at the new shapes.  The file sorts last so that a failure here cannot mask the separate()-path tests under -x.

Tolerances: fp32 mode 1e-3 max-abs (the north-star bound); bf16 mode (bf16 GEMM operands, fp32 accumulation /
residual stream / norms) against stated looser bounds, measured error printed.
"""
import pytest
import torch

from oracle import gen_golden_judge as G
from oracle import judge_oracle as J
from oracle import samaudio_oracle as O
from sam_audio_amd import hip
from sam_audio_amd.config import PEAudioFrameConfig, PEAVTransformerConfig, SAMAudioJudgeConfig
from sam_audio_amd.synthetic import init_frame_state_dict, init_judge_state_dict, synthetic_clip
from tests import util

pytestmark = pytest.mark.gpu
# bf16 bounds = 2 x measured on MI355X (profiles/r2_gpu_tests_call2_measured_errors.log): PE-AV hidden 1.07e-2 on |x| <= 2.4
TOL = {"fp32": 1e-3, "bf16": 2.5e-2}
TINY_TEXT = dict(G.TINY_TEXT)


def _cfg(codec=None) -> SAMAudioJudgeConfig:
    """Tiny transformers on the DEFAULT codec dims (the codec kernels are exercised at the shapes they ship with)."""
    return SAMAudioJudgeConfig(audio_codec=codec, transformer=G.TINY_TC, finetune_transformer=G.TINY_FT,
                               text_model=TINY_TEXT, nth_text_layer=2, bottleneck_dim=64)


# ---------------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_masked_groupnorm_silu(gpu, prec):
    B, S, Cc, halo = 3, 37, 256, 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, S, Cc, generator=g) * 1.5 + 0.3
    w, b = torch.randn(Cc, generator=g) * 0.2 + 1, torch.randn(Cc, generator=g) * 0.1
    mask = torch.arange(S)[None] < torch.tensor([37, 20, 1])[:, None]
    want = torch.nn.functional.silu(J.masked_group_norm_1(x, mask, w, b))
    out = torch.full((B, S + 2 * halo, Cc), float("nan"), dtype=util.ACT_DT[prec], device=gpu)
    part = torch.empty(B * 64 * 3, dtype=torch.float64, device=gpu)
    xd, wd, bd, md = x.to(gpu), w.to(gpu), b.to(gpu), mask.to(gpu).to(torch.uint8)   # kept alive across the launch
    hip.check(hip.lib().samaudio_op_masked_groupnorm_silu(
        hip.ptr(xd), hip.ptr(wd), hip.ptr(bd), hip.ptr(md), hip.ptr(part), hip.ptr(out), util.PREC[prec], B, S, Cc,
        halo, 1e-5, util.stream()))
    util.report(f"masked groupnorm {prec}", out[:, halo:halo + S], want, 1e-4 if prec == "fp32" else 3e-2)
    assert torch.isnan(out[:, 0].float()).all() and torch.isnan(out[:, -1].float()).all(), "halo rows were touched"


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_layernorm_rows(gpu, prec):
    M, D = 9, 320
    g = torch.Generator().manual_seed(4)
    x, w, b = torch.randn(M, D, generator=g) * 2 + 1, torch.randn(D, generator=g), torch.randn(D, generator=g)
    want = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-6)
    o32 = torch.empty(M, D, device=gpu)
    oact = torch.empty(M, D, dtype=util.ACT_DT[prec], device=gpu)
    xd, wd, bd = x.to(gpu), w.to(gpu), b.to(gpu)
    hip.check(hip.lib().samaudio_op_layernorm_rows(hip.ptr(xd), D, hip.ptr(wd), hip.ptr(bd), hip.ptr(o32),
                                                   hip.ptr(oact), util.PREC[prec], M, D, 1e-6, util.stream()))
    util.report("layernorm rows f32 out", o32, want, 1e-4)
    util.report(f"layernorm rows act out {prec}", oact, want, 1e-4 if prec == "fp32" else 3e-2)


# ---------------------------------------------------------------------------------------------------- transformer
def _judge(cfg, sd, prec, gpu, text_model=None):
    from sam_audio_amd.judge import SAMAudioJudgeModel
    m = SAMAudioJudgeModel(cfg, precision=prec, device=str(gpu), text_model=text_model or G.text_tower(cfg))
    m.load_state_dict(sd, strict=False)
    return m


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("masked", [True, False])
def test_peav_transformer_matches_oracle(gpu, prec, masked):
    """One PE-AV transformer (input projection, class token, masked-GroupNorm ResNet block, layers, output
    projection) through the samaudio_judge_encode hook against oracle peav_transformer, which tests/test_judge_oracle.py
    pins to the Hugging Face PeAudioEncoder on this very network (same seeds as gen_golden_judge.pin_transformer)."""
    from sam_audio_amd.judge import _ensure_ws
    from sam_audio_amd.synthetic import init_peav_state_dict
    tc = PEAVTransformerConfig(**G.TINY_TC)
    g = torch.Generator().manual_seed(5)
    psd = init_peav_state_dict(tc, "transformer.", g, torch.device("cpu"))
    mask = torch.arange(21)[None] < torch.tensor([21, 13, 6])[:, None]
    cfg = _cfg(codec=dict(codebook_dim=64))
    sd = init_judge_state_dict(cfg, seed=9, with_codec=False)
    sd.update(psd)
    z = torch.randn(3, 21, 64, generator=g)                                  # codec features -> data_proj -> layers
    xin = torch.nn.functional.linear(z, sd["data_proj.weight"], sd["data_proj.bias"])
    with torch.inference_mode():
        last, pooled = J.peav_transformer(sd, "transformer.", xin, mask if masked else None,
                                          n_heads=tc.num_attention_heads, n_layers=tc.num_hidden_layers,
                                          eps=tc.rms_norm_eps, rope_theta=tc.rope_theta)
    m = _judge(cfg, sd, prec, gpu)
    hidden = torch.empty(3, 22, tc.hidden_size, device=gpu)
    pm = mask.to(gpu).to(torch.uint8).contiguous() if masked else None
    need = m._lib.samaudio_judge_workspace_bytes(m._h, 3, 1, 21)
    _ensure_ws(m, need, lambda p, n: m._lib.samaudio_judge_set_workspace(m._h, p, n))
    zd = z.to(gpu).contiguous()
    hip.check(m._lib.samaudio_judge_encode(m._h, 0, hip.ptr(zd), hip.ptr(pm), 3, 21, hip.ptr(hidden), util.stream()))
    valid = (mask if masked else torch.ones_like(mask))[..., None]
    util.report(f"peav pooled {prec}", hidden[:, 0], pooled, TOL[prec])
    util.report(f"peav last_hidden {prec}", hidden[:, 1:].cpu() * valid, last * valid, TOL[prec])


# ---------------------------------------------------------------------------------------------------- Judge
def _judge_case(cfg, B=2, T=6, cand=1, seed=21, ragged=True):
    g = torch.Generator().manual_seed(seed)
    hop = cfg.audio_codec.hop_length
    lengths = torch.tensor([T * hop] + [max(2, T - 2 - i) * hop for i in range(B - 1)]) if ragged else torch.full((B,), T * hop)
    pad = torch.arange(T * hop)[None] < lengths[:, None]
    wav_in = torch.stack([synthetic_clip(i, T * hop) for i in range(B)]) * pad[:, None]
    wav_sep = 0.5 * torch.stack([synthetic_clip(10 + i, T * hop) for i in range(B * cand)]) * pad.repeat_interleave(cand, 0)[:, None]
    ids = torch.randint(3, 128, (B, 6), generator=g)
    ids[:, 0] = 1
    att = torch.ones(B, 6, dtype=torch.long)
    att[-1, 4:] = 0
    return dict(input_ids=ids, attention_mask=att, input_values=wav_in, separated_values=wav_sep, padding_mask=pad)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_judge_forward_matches_oracle(gpu, prec):
    cfg = _cfg()
    sd = init_judge_state_dict(cfg, seed=9)
    inp = _judge_case(cfg)
    tm = G.text_tower(cfg)
    pooled = G.text_pooled(tm, cfg, inp["input_ids"], inp["attention_mask"])
    with torch.inference_mode():
        want = J.judge_forward(sd, cfg, pooled, inp["input_values"], inp["separated_values"], inp["padding_mask"])
    m = _judge(cfg, sd, prec, gpu, text_model=tm)
    out = m(**{k: v.to(gpu) for k, v in inp.items()})
    got = torch.cat([out.overall, out.recall, out.precision, out.faithfulness], dim=1)
    util.report(f"judge scores {prec}", got, want, 1e-3 if prec == "fp32" else 1e-2)   # measured 3.5e-3
    assert out.overall.shape == (2, 1)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_judge_candidate_dedup_equals_the_expanded_batch(gpu, prec):
    """score_candidates (mixture branch once per clip) == forward() on the reference's expanded batch
    (ranking/judge.py:31-33), bit for bit within a precision mode, and == the oracle."""
    cfg = _cfg()
    sd = init_judge_state_dict(cfg, seed=9)
    cand = 3
    inp = _judge_case(cfg, B=2, T=5, cand=cand)
    tm = G.text_tower(cfg)
    pooled = G.text_pooled(tm, cfg, inp["input_ids"], inp["attention_mask"]).repeat_interleave(cand, 0)  # on the CPU,
    m = _judge(cfg, sd, prec, gpu, text_model=tm)                              # before the tower moves to the GPU
    scores = m.score_candidates(inp["input_ids"].to(gpu), inp["input_values"].to(gpu), inp["separated_values"].to(gpu),
                                cand, attention_mask=inp["attention_mask"].to(gpu), padding_mask=inp["padding_mask"].to(gpu))
    expanded = m(input_ids=inp["input_ids"].repeat_interleave(cand, 0).to(gpu),
                 attention_mask=inp["attention_mask"].repeat_interleave(cand, 0).to(gpu),
                 input_values=inp["input_values"].repeat_interleave(cand, 0).to(gpu),
                 separated_values=inp["separated_values"].to(gpu),
                 padding_mask=inp["padding_mask"].repeat_interleave(cand, 0).to(gpu))
    assert scores.shape == (2, cand)
    util.report(f"dedup vs expanded {prec}", scores.reshape(-1, 1), expanded.overall.cpu(), 1e-5 if prec == "fp32" else 2e-2)
    with torch.inference_mode():
        want = J.judge_forward(sd, cfg, pooled, inp["input_values"].repeat_interleave(cand, 0), inp["separated_values"],
                               inp["padding_mask"].repeat_interleave(cand, 0))
    util.report(f"dedup vs oracle {prec}", scores.reshape(-1), want[:, 0], 1e-3 if prec == "fp32" else 1e-2)   # measured 2.2e-3


def test_separate_with_judge_reranking_picks_the_argmax(gpu):
    """separate(reranking_candidates=2) end to end with the HIP Judge as text_ranker: the returned target is the
    candidate the Judge scores highest (reference model.py:318-328), checked against scores recomputed here."""
    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.processor import SAMAudioJudgeProcessor
    from sam_audio_amd.ranking import JudgeRanker
    from sam_audio_amd.synthetic import init_state_dict, synthetic_noise, synthetic_text_features
    from tests.test_judge_host_cpu import _Tok
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=3)
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 6 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 4)
    batch = SAMAudioProcessor.from_config(cfg)(descriptions=["dog", "rain"], audios=clips, text_features=text,
                                               text_mask=tmask).to(gpu)
    model = SAMAudio(cfg, precision="fp32", device=str(gpu))
    model.load_state_dict(sd)
    jcfg = _cfg()
    judge = _judge(jcfg, init_judge_state_dict(jcfg, seed=9), "fp32", gpu)
    ranker = JudgeRanker(model=judge, processor=SAMAudioJudgeProcessor(hop, 48000, tokenizer=_Tok()))
    seen = {}

    def spy(**kw):
        seen["scores"] = ranker(**kw)
        return seen["scores"]

    model.text_ranker = spy
    noise = synthetic_noise(4, 6)
    res = model.separate(batch, noise=noise.to(gpu), reranking_candidates=2)
    assert seen["scores"].shape == (2, 2)
    pick = seen["scores"].argmax(dim=1).tolist()
    model.text_ranker = None
    first = model.separate(batch, noise=noise.to(gpu), reranking_candidates=2)   # candidate 0 of each clip
    lat = model.last_latent.view(2, 2, 6, -1)
    for b in range(2):
        half = lat.shape[-1] // 2
        want = model.decode_audio(lat[b, pick[b], :, :half][None].contiguous())[0]
        assert torch.allclose(res.target[b], want[: res.target[b].numel()], atol=1e-5)
        if pick[b] == 0:
            assert torch.equal(res.target[b], first.target[b])


# ---------------------------------------------------------------------------------------------------- PE-A-Frame
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_frame_logits_and_spans_match_oracle(gpu, prec):
    from sam_audio_amd.judge import PEAudioFrame
    cfg = PEAudioFrameConfig(audio=G.TINY_TC, text_model=dict(TINY_TEXT, hidden_size=64), codebook_dim=128)
    sd = init_frame_state_dict(cfg, seed=2)
    g = torch.Generator().manual_seed(6)
    B, T = 3, 50
    feats = torch.randn(B, T, 128, generator=g)
    pooled = torch.randn(B, cfg.text_hidden, generator=g)
    pad = torch.arange(T)[None] < torch.tensor([50, 31, 9])[:, None]
    with torch.inference_mode():
        want = J.frame_logits(sd, cfg, pooled, feats, pad)
    import transformers
    torch.manual_seed(1)
    tm = transformers.ModernBertModel(transformers.ModernBertConfig(**cfg.text_model)).eval()
    fp = PEAudioFrame(cfg, precision=prec, device=str(gpu), text_model=tm)
    fp.load_state_dict(sd, strict=False)
    out = fp(input_features=feats.to(gpu), padding_mask=pad.to(gpu), return_spans=True, text_pooled=pooled.to(gpu))
    scale = max(1.0, want.abs().max().item())
    util.report(f"frame logits {prec}", out.logits.cpu() * pad, want * pad, (1e-3 if prec == "fp32" else 1e-2) * scale)   # bf16 measured 4.4e-3 * scale
    if prec == "fp32":
        margin = (want.abs() > 1e-2) | ~pad                                     # frames not sitting on the threshold
        ids_w, al_w = O.anchors_to_ids([[("+", s, e) for s, e in r] for r in J.spans_from_logits(want, pad, 1920, 48000)],
                                       pad, 1920, 48000)
        ids_g, al_g = O.anchors_to_ids([[("+", s, e) for s, e in r] for r in out.spans], pad, 1920, 48000)
        assert torch.equal((al_w >= 2) & margin, (al_g >= 2) & margin), "span frames differ away from the threshold"


def test_predict_spans_reproduces_quirk_q13_and_the_opt_in_fix(gpu):
    """reference model.py:257 builds the forward args BEFORE predict_spans rebinds the batch's anchor tensors
    (:259-268, processor.py:122-123), so in the reference snapshot predicted spans never reach the ODE: the output
    equals predict_spans=False while batch.anchors is filled in.  fix_span_order=True feeds them to the ODE, which must
    equal a run with the same anchors given explicitly."""
    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.judge import PEAudioFrameOutput
    from sam_audio_amd.synthetic import init_state_dict, synthetic_noise, synthetic_text_features
    cfg = preset_config("tiny")
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 5 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 3)
    proc = SAMAudioProcessor.from_config(cfg)
    spans = [[[0.04 - 1e-6, 0.12 - 1e-6]], [[0.0, 0.08 - 1e-6], [0.16 - 1e-6, 0.2 - 1e-6]]]
    seen = {}

    def predictor(input_features, padding_mask=None, return_spans=False, **kw):
        seen.update(shape=tuple(input_features.shape), return_spans=return_spans, keys=sorted(kw))
        return PEAudioFrameOutput(logits=torch.zeros(input_features.shape[:2]), spans=spans)

    model = SAMAudio(cfg, precision="fp32", device=str(gpu))
    model.load_state_dict(init_state_dict(cfg, seed=3))
    noise = synthetic_noise(2, 5).to(gpu)

    def run(**kw):
        batch = proc(descriptions=["a", "b"], audios=clips, text_features=text, text_mask=tmask,
                     anchors=kw.pop("anchors", None)).to(gpu)
        res = model.separate(batch, noise=noise, **kw)
        return batch, model.last_latent.clone(), res

    _, base, _ = run()
    model.span_predictor = predictor
    model.span_predictor_transform = lambda text: {"input_ids": torch.ones(len(text), 2, dtype=torch.long), "extra": 1}
    batch, q13, _ = run(predict_spans=True)
    assert seen["shape"] == (2, 5, 128) and seen["return_spans"] and seen["keys"] == ["input_ids"]
    assert batch.anchors == [[("+", s, e) for s, e in row] for row in spans] and int(batch.anchor_alignment.max()) >= 2
    assert torch.equal(q13, base), "quirk Q13: predicted spans must not reach the ODE by default"
    model.fix_span_order = True
    _, fixed, _ = run(predict_spans=True)
    model.span_predictor = None
    _, explicit, _ = run(anchors=[[("+", s, e) for s, e in row] for row in spans])
    assert torch.equal(fixed, explicit) and not torch.equal(fixed, base)


def test_visual_prompt_features_reach_the_ode(gpu):
    """Row a4: separate() with masked videos - processor frame sampling (processor.py:147-153) -> PerceptionEncoder
    wrapper (transform, chunking, time padding; vision_encoder.py:47-113) around an injected tower -> the video term of
    align_inputs (align.py:41-50) inside the HIP prepare step, against the oracle fed with the same features."""
    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.config import PerceptionEncoderConfig
    from sam_audio_amd.synthetic import init_state_dict, synthetic_noise, synthetic_text_features
    from sam_audio_amd.vision_encoder import PerceptionEncoder
    cfg = preset_config("tiny")
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 4 * hop) for i in range(2)]
    text, tmask = synthetic_text_features(2, 3)
    g = torch.Generator().manual_seed(12)
    videos = [torch.randint(0, 256, (7, 3, 10, 12), generator=g, dtype=torch.uint8),
              torch.randint(0, 256, (5, 3, 8, 8), generator=g, dtype=torch.uint8)]
    wproj = torch.randn(3, cfg.vision_encoder.dim, generator=g)

    def tower(frames, normalize=True):
        f = frames.float().mean(dim=(2, 3)).cpu() @ wproj                       # [N, 1024]
        return torch.nn.functional.normalize(f, dim=-1) if normalize else f

    enc = PerceptionEncoder(PerceptionEncoderConfig(image_size=8, batch_size=3), tower)
    proc = SAMAudioProcessor.from_config(cfg)
    batch = proc(descriptions=["a", "b"], audios=clips, masked_videos=videos, text_features=text, text_mask=tmask)
    assert [tuple(v.shape) for v in batch.masked_video] == [(4, 3, 10, 12), (4, 3, 8, 8)]   # one frame per latent step
    feats_v = enc(batch.masked_video)                                            # [B, T, 1024]
    noise = synthetic_noise(2, 4)
    sd = init_state_dict(cfg, seed=3)
    with torch.inference_mode():
        _, _, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise,
                                   video=feats_v.transpose(1, 2), decode=False)
        _, _, lat_novid = O.separate(sd, cfg, batch.audios, batch.sizes.long(), text, tmask, noise, decode=False)
    assert (lat_ref - lat_novid).abs().max() > 1e-3, "the video term must matter for this check to mean anything"
    model = SAMAudio(cfg, precision="fp32", device=str(gpu))
    model.load_state_dict(sd)
    model.vision_encoder = enc
    model.separate(batch.to(gpu), noise=noise.to(gpu))
    util.report("latent with visual prompt", model.last_latent, lat_ref, 1e-3)
    # with reranking candidates the video features are repeated sample-major like every other conditioning tensor
    # (reference model.py:193-229)
    noise2 = synthetic_noise(4, 4, seed=5)
    with torch.inference_mode():
        _, _, lat2 = O.separate(sd, cfg, batch.audios.cpu(), batch.sizes.long().cpu(), text, tmask, noise2,
                                video=feats_v.transpose(1, 2), candidates=2, decode=False)
    model.separate(batch, noise=noise2.to(gpu), reranking_candidates=2)
    util.report("latent with visual prompt, 2 candidates", model.last_latent, lat2, 1e-3)


def test_t5_text_encoder_on_the_gpu_feeds_separate(gpu):
    """Row a3 on the device: descriptions -> `T5TextEncoder` (the T5 stack on the HIP library, weights of a random-init
    transformers T5EncoderModel at the model's text width, whitespace-hash tokenizer - t5-base's files cannot be fetched
    offline; parity of the stack itself: tests/test_t5_gpu.py) -> `[B, Lt, 768]`
    features + bool mask -> the HIP prepare step (reference text_encoder.py:19-37, model.py:256-257).  separate() through
    `model.text_encoder` must equal separate() fed with the same features explicitly (bitwise), and the oracle (1e-3)."""
    import transformers
    from sam_audio_amd import SAMAudio, SAMAudioProcessor, preset_config
    from sam_audio_amd.synthetic import init_state_dict, synthetic_noise
    from sam_audio_amd.text_encoder import T5TextEncoder
    if gpu.type != "cuda":
        pytest.skip("needs PyTorch-ROCm on a GPU (the dry-run builds replace torch's device plumbing)")

    class Tok:
        def __call__(self, texts, truncation=True, max_length=512, padding="longest", return_tensors="pt"):
            rows = [[2 + (sum(map(ord, w)) % 90) for w in t.split()][: max_length - 1] + [1] for t in texts]
            width = max(len(r) for r in rows)
            ids = torch.zeros(len(rows), width, dtype=torch.long)
            att = torch.zeros(len(rows), width, dtype=torch.long)
            for i, r in enumerate(rows):
                ids[i, : len(r)] = torch.tensor(r)
                att[i, : len(r)] = 1
            return {"input_ids": ids, "attention_mask": att}

    cfg = preset_config("tiny")
    torch.manual_seed(4)
    t5 = transformers.T5EncoderModel(transformers.T5Config(vocab_size=100, d_model=cfg.text_encoder.dim, d_kv=32, d_ff=256,
                                                           num_layers=2, num_heads=4))
    enc = T5TextEncoder(cfg.text_encoder, model=t5, tokenizer=Tok(), device=gpu)
    descriptions = ["a dog barking loudly", "rain"]
    feats, mask = enc(descriptions)
    assert feats.shape == (2, 5, cfg.text_encoder.dim) and feats.device.type == "cuda"
    assert mask.tolist() == [[True] * 5, [True, True, False, False, False]]
    hop = cfg.audio_codec.hop_length
    clips = [synthetic_clip(i, 5 * hop) for i in range(2)]
    proc = SAMAudioProcessor.from_config(cfg)
    noise = synthetic_noise(2, 5)
    sd = init_state_dict(cfg, seed=3)
    model = SAMAudio(cfg, precision="fp32", device=str(gpu), text_encoder=enc)
    model.load_state_dict(sd)
    res_a = model.separate(proc(descriptions=descriptions, audios=clips).to(gpu), noise=noise.to(gpu))
    lat_a = model.last_latent.clone()
    res_b = model.separate(proc(descriptions=descriptions, audios=clips, text_features=feats, text_mask=mask).to(gpu),
                           noise=noise.to(gpu))
    assert torch.equal(lat_a, model.last_latent) and all(torch.equal(x, y) for x, y in zip(res_a.target, res_b.target))
    batch = proc(descriptions=descriptions, audios=clips)
    with torch.inference_mode():
        _, _, lat_ref = O.separate(sd, cfg, batch.audios, batch.sizes.long(), feats.cpu(), mask.cpu(), noise, decode=False)
    util.report("latent with T5 features from the GPU", lat_a, lat_ref, 1e-3)
