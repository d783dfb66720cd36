"""Host logic of the reranking / span-prediction rows, no GPU needed: weight conversion against the C ABI's shape
checks, the processors, ranker plumbing, candidate selection and the T5 wrapper."""
import ctypes as C

import pytest
import torch

from oracle import gen_golden_judge as G
from oracle import judge_oracle as J
from sam_audio_amd import hip
from sam_audio_amd.config import JudgeRankerConfig, PEAudioFrameConfig, SAMAudioJudgeConfig, T5EncoderConfig
from sam_audio_amd.judge import (JUDGE_OWN_KEYS, convert_frame, convert_judge, peav_dims, peav_keys,
                                 spans_from_logits)
from sam_audio_amd.processor import SAMAudioJudgeProcessor
from sam_audio_amd.ranking import EnsembleRanker, JudgeRanker, Ranker, create_ranker
from sam_audio_amd.synthetic import init_frame_state_dict, init_judge_state_dict


def _judge_handle(cfg, prec):
    jc = hip.JudgeConfig(precision=prec, transformer=peav_dims(cfg.transformer, cfg.audio_codec.codebook_dim),
                         finetune_transformer=peav_dims(cfg.finetune_transformer, cfg.bottleneck_dim),
                         codec_dim=cfg.audio_codec.codebook_dim, text_hidden=cfg.text_hidden,
                         bottleneck_dim=cfg.bottleneck_dim)
    h = C.c_void_p()
    hip.check(hip.lib().samaudio_judge_create(C.byref(jc), C.byref(h)))
    return h


def _set_all(fn, h, tensors):
    for name, t in tensors.items():
        dt = {torch.float32: hip.DT_F32, torch.bfloat16: hip.DT_BF16}[t.dtype]
        hip.check(fn(h, name.encode(), hip.ptr(t), dt, t.dim(), hip.shape_array(t.shape)))


@pytest.mark.parametrize("prec,dtype", [(hip.F32, torch.float32), (hip.BF16, torch.bfloat16)])
def test_judge_context_accepts_converted_weights_and_reports_missing_ones(prec, dtype):
    """set_tensor / finalize look at names, dtypes and shapes only, so CPU pointers are fine here."""
    cfg = G.tiny_judge_config()
    sd = init_judge_state_dict(cfg, seed=1)
    tensors = convert_judge(sd, cfg, dtype, "cpu")
    lib = hip.lib()
    h = _judge_handle(cfg, prec)
    try:
        _set_all(lib.samaudio_judge_set_tensor, h, {k: v for k, v in tensors.items() if k != "ft.L0.w13"})
        with pytest.raises(RuntimeError, match="ft.L0.w13"):
            hip.check(lib.samaudio_judge_finalize(h))
        _set_all(lib.samaudio_judge_set_tensor, h, {"ft.L0.w13": tensors["ft.L0.w13"]})
        hip.check(lib.samaudio_judge_finalize(h))
        small = lib.samaudio_judge_workspace_bytes(h, 2, 1, 12)
        big = lib.samaudio_judge_workspace_bytes(h, 2, 4, 12)
        assert 0 < small < big
        with pytest.raises(hip.SamAudioHipError):  # no workspace yet
            hip.check(lib.samaudio_judge_score(h, hip.ptr(torch.zeros(4)), hip.ptr(torch.zeros(4)), 1, 1, 4,
                                               hip.ptr(torch.zeros(4)), None, hip.ptr(torch.zeros(4)), None))
    finally:
        lib.samaudio_judge_destroy(h)


def test_judge_conversion_layout():
    cfg = G.tiny_judge_config()
    sd = init_judge_state_dict(cfg, seed=1)
    t = convert_judge(sd, cfg, torch.float32, "cpu")
    D, Bn = cfg.transformer.hidden_size, cfg.bottleneck_dim
    x = torch.randn(3, 2 * D)
    ref = torch.nn.functional.linear(x, sd["cat_audio_proj.weight"], sd["cat_audio_proj.bias"])
    mine = x[:, :D] @ t["cat.wh"].T + x[:, D:] @ t["cat.wi"].T + t["cat.b"]      # [hyp | input] halves
    assert torch.allclose(ref, mine, atol=1e-5)
    y = torch.randn(3, 2 * Bn)
    ref = torch.nn.functional.linear(y, sd["proj_audio_and_text.weight"], sd["proj_audio_and_text.bias"])
    assert torch.allclose(ref, y[:, :Bn] @ t["pat.wa"].T + y[:, Bn:] @ t["pat.wt"].T + t["pat.b"], atol=1e-5)
    q = sd["transformer.layers.1.self_attn.q_proj.weight"]
    assert torch.equal(t["t.L1.wqkv"][:D], q) and t["t.L1.wqkv"].shape == (3 * D, D)
    w13 = t["t.L0.w13"]                                                          # 16-row gate/up interleave
    assert torch.equal(w13[0:16], sd["transformer.layers.0.mlp.gate_proj.weight"][0:16])
    assert torch.equal(w13[16:32], sd["transformer.layers.0.mlp.up_proj.weight"][0:16])
    conv = sd["transformer.patch_embedder.resnet_block.block1.project.weight"]   # [Cout, Cin, 3] -> tap-major
    assert torch.equal(t["t.conv1.w"][:, D:2 * D], conv[:, :, 1])
    assert t["t.rope_cos"].shape == (cfg.transformer.max_position_embeddings, 64)
    assert set(JUDGE_OWN_KEYS) <= set(sd) and set(peav_keys(cfg.transformer, "transformer.")) <= set(sd)


def test_frame_context_accepts_converted_weights():
    cfg = PEAudioFrameConfig(audio=G.TINY_TC, text_model=dict(G.TINY_TEXT, hidden_size=64), codebook_dim=64)
    sd = init_frame_state_dict(cfg, seed=2)
    tensors = convert_frame(sd, cfg, torch.bfloat16, "cpu")
    lib = hip.lib()
    fc = hip.FrameConfig(precision=hip.BF16, audio=peav_dims(cfg.audio, cfg.codebook_dim), codec_dim=cfg.codebook_dim,
                         embed_dim=cfg.text_hidden)
    h = C.c_void_p()
    hip.check(lib.samaudio_frame_create(C.byref(fc), C.byref(h)))
    try:
        with pytest.raises(RuntimeError, match="missing weight"):
            hip.check(lib.samaudio_frame_finalize(h))
        _set_all(lib.samaudio_frame_set_tensor, h, tensors)
        hip.check(lib.samaudio_frame_finalize(h))
        assert lib.samaudio_frame_workspace_bytes(h, 2, 30) > 0
    finally:
        lib.samaudio_frame_destroy(h)


def test_unsupported_judge_configs_are_rejected():
    with pytest.raises(NotImplementedError):
        SAMAudioJudgeConfig(transformer=dict(hidden_size=192, num_attention_heads=3, head_dim=64)).check_supported()
    with pytest.raises(NotImplementedError):
        SAMAudioJudgeConfig(transformer=dict(hidden_act="gelu")).check_supported()
    SAMAudioJudgeConfig().check_supported()  # pe-av-large defaults: 1792 = 14 x 128


class _Tok:
    def __call__(self, text, return_tensors="pt", padding="longest", max_length=512, truncation=True):
        rows = [[1] + [3 + (ord(c) % 100) for c in t.split()[0][:5]] for t in text]
        width = max(len(r) for r in rows)
        ids = torch.zeros(len(rows), width, dtype=torch.long)
        att = torch.zeros(len(rows), width, dtype=torch.long)
        for i, r in enumerate(rows):
            ids[i, :len(r)] = torch.tensor(r)
            att[i, :len(r)] = 1
        return {"input_ids": ids, "attention_mask": att}


def test_judge_processor_pads_to_the_hop_and_masks():
    proc = SAMAudioJudgeProcessor(16, 48000, tokenizer=_Tok())
    a, b = torch.randn(1, 40), torch.randn(1, 64)
    out = proc(text=["dog barking", "rain"], input_audio=[a, b], separated_audio=[a * 0.5, b * 0.5], sampling_rate=48000)
    assert out["input_values"].shape == (2, 1, 64) and out["separated_values"].shape == (2, 1, 64)
    assert out["padding_mask"].sum(1).tolist() == [48, 64]                        # 40 -> reflect-padded to 48
    assert torch.equal(out["input_values"][0, 0, :40], a[0])
    assert torch.equal(out["input_values"][0, 0, 40:48], a[0].flip(0)[1:9])       # right reflect pad
    assert out["input_ids"].shape[0] == 2 and out["attention_mask"][0].sum() < out["attention_mask"][1].sum()
    with pytest.raises(ValueError):
        proc(input_audio=[a], sampling_rate=16000)
    with pytest.raises(ValueError):
        proc(input_audio=["clip.wav"])
    assert out.to("cpu")["input_values"].shape == (2, 1, 64)


class _Const(Ranker):
    def __init__(self, scores):
        self.scores = scores

    def forward(self, **kwargs):
        return self.scores


def test_ensemble_and_create_ranker():
    a, b = torch.tensor([[1.0, 0.0]]), torch.tensor([[0.0, 3.0]])
    ens = EnsembleRanker([_Const(a), _Const(b)], [1.0, 0.5])
    assert torch.equal(ens(extracted_audio=None), torch.tensor([[1.0, 1.5]]))
    assert create_ranker(None) is None and create_ranker(ens) is ens
    with pytest.raises(NotImplementedError):
        create_ranker({"kind": "clap"})
    with pytest.raises(FileNotFoundError):
        create_ranker(JudgeRankerConfig("facebook/sam-audio-judge"))              # hub ids are unreachable offline


def test_judge_ranker_hands_each_mixture_over_once():
    """ranking/judge.py:29-42 shapes: B clips x candidates; the mixture goes in once per clip."""
    seen = {}

    class FakeJudge:
        def score_candidates(self, input_ids, input_values, separated_values, candidates, attention_mask=None,
                             padding_mask=None):
            seen.update(ids=input_ids, inp=input_values, sep=separated_values, cand=candidates, pad=padding_mask)
            return torch.arange(input_values.shape[0] * candidates, dtype=torch.float32).view(-1, candidates)

    r = JudgeRanker(model=FakeJudge(), processor=SAMAudioJudgeProcessor(16, 48000, tokenizer=_Tok()))
    mix = [torch.randn(1, 48).expand(3, -1), torch.randn(1, 32).expand(3, -1)]
    ext = [torch.randn(3, 48), torch.randn(3, 32)]
    scores = r(input_audio=mix, extracted_audio=ext, descriptions=["a", "b"], sample_rate=48000)
    assert scores.shape == (2, 3) and seen["cand"] == 3
    assert seen["inp"].shape == (2, 1, 48) and seen["sep"].shape == (6, 1, 48) and seen["ids"].shape[0] == 2
    assert torch.equal(seen["sep"][4, 0, :32], ext[1][1]) and seen["pad"].sum(1).tolist() == [48, 32]


def test_candidate_selection_follows_the_reference_order():
    """model.py:306-330: visual ranker only with a masked video, else the text ranker, else candidate 0."""
    from sam_audio_amd import preset_config
    from sam_audio_amd.model import SAMAudio
    m = SAMAudio(preset_config("tiny"), precision="fp32", device="cpu")
    calls = []

    class Rk(Ranker):
        def __init__(self, name, scores):
            self.name, self.scores = name, scores

        def forward(self, **kw):
            calls.append((self.name, sorted(kw)))
            return self.scores

    class B:
        audios = torch.randn(2, 1, 64)
        descriptions = ["a", "b"]
        masked_video = None

    tgt = [torch.randn(3, 64), torch.randn(3, 32)]
    sizes = torch.tensor([64, 32])
    assert m._rerank(B, tgt, sizes, 3).tolist() == [0, 0]
    m.text_ranker = Rk("text", torch.tensor([[0.1, 0.9, 0.2], [0.5, 0.1, 0.7]]))
    m.visual_ranker = Rk("visual", torch.tensor([[0.9, 0.1, 0.2], [0.5, 0.8, 0.7]]))
    assert m._rerank(B, tgt, sizes, 3).tolist() == [1, 2]
    assert calls[-1] == ("text", ["descriptions", "extracted_audio", "input_audio", "sample_rate"])
    assert m._rerank(B, tgt, sizes, 1).tolist() == [0, 0]
    B.masked_video = [torch.zeros(1)]
    assert m._rerank(B, tgt, sizes, 3).tolist() == [0, 1] and calls[-1][0] == "visual"


def test_product_span_rule_equals_the_oracle_rule():
    g = torch.Generator().manual_seed(8)
    logits = torch.randn(3, 40, generator=g) * 2
    pad = torch.arange(40)[None] < torch.tensor([40, 25, 7])[:, None]
    for th in (0.5, 0.3):
        assert spans_from_logits(logits, pad, 1920, 48000, th) == J.spans_from_logits(logits, pad, 1920, 48000, th)


def test_t5_text_encoder_wrapper_has_no_cpu_execution_path():
    """The wrapper validates its container on the CPU, but only the HIP stack ever runs it (tests/test_t5_gpu.py compares that
    stack with the transformers module)."""
    transformers = pytest.importorskip("transformers")
    from sam_audio_amd import hip
    from sam_audio_amd.text_encoder import T5TextEncoder
    torch.manual_seed(0)
    t5 = transformers.T5EncoderModel(transformers.T5Config(d_model=64, d_kv=16, d_ff=128, num_layers=2, num_heads=4,
                                                           vocab_size=128)).eval()

    class Tok:
        def __call__(self, texts, truncation=True, max_length=512, padding="longest", return_tensors="pt"):
            return _Tok()(texts)

    enc = T5TextEncoder(T5EncoderConfig(name="unused", dim=64), model=t5, tokenizer=Tok())
    assert enc.backend == "hip" and enc.device is None
    with pytest.raises(hip.SamAudioHipError, match="no CPU fallback"):
        enc(["dog barking", "rain"])
    with pytest.raises(hip.SamAudioHipError, match="no CPU fallback"):
        enc.to("cpu")
    with pytest.raises(ValueError):
        T5TextEncoder(T5EncoderConfig(name="unused", dim=768), model=t5, tokenizer=Tok())
    with pytest.raises(FileNotFoundError):
        T5TextEncoder(T5EncoderConfig(name="t5-base"))                           # nothing cached offline


def test_perception_encoder_wrapper_transform_chunking_and_padding():
    from sam_audio_amd.config import PerceptionEncoderConfig
    from sam_audio_amd.vision_encoder import PerceptionEncoder
    calls = []

    def tower(frames, normalize=True):
        calls.append((tuple(frames.shape), normalize))
        f = frames.mean(dim=(2, 3))                                   # [N, 3]
        return torch.cat([f, f.new_ones(f.shape[0], 5)], dim=1)

    enc = PerceptionEncoder(PerceptionEncoderConfig(dim=8, batch_size=4, image_size=6), tower)
    vids = [torch.full((10, 3, 6, 6), 255, dtype=torch.uint8), torch.zeros(3, 3, 12, 9, dtype=torch.uint8)]
    out = enc(vids)
    assert out.shape == (2, 10, 8)
    assert [c[0][0] for c in calls] == [4, 4, 2, 3] and all(c[1] for c in calls)      # chunks of batch_size
    assert torch.allclose(out[0, :, :3], torch.ones(10, 3)) and torch.allclose(out[1, :3, :3], -torch.ones(3, 3))
    assert torch.equal(out[1, 3:], torch.zeros(7, 8))                                   # time padding
    with pytest.raises(ValueError):
        PerceptionEncoder(PerceptionEncoderConfig(interpolation_mode="lanczos"))
    with pytest.raises(NotImplementedError):                                              # no tower for an unknown config name
        PerceptionEncoder(PerceptionEncoderConfig(name="PE-Core-unknown", image_size=6))([vids[0]])
    with pytest.raises(ValueError):                                                       # known name, inconsistent geometry
        PerceptionEncoder(PerceptionEncoderConfig(image_size=6))


def test_attach_rankers_skips_what_cannot_be_built_offline():
    import warnings
    from sam_audio_amd import preset_config
    from sam_audio_amd.model import SAMAudio
    cfg = preset_config("tiny", text_ranker={"kind": "judge", "checkpoint_or_model_id": "facebook/sam-audio-judge"},
                        visual_ranker={"kind": "imagebind", "checkpoint": None})
    m = SAMAudio(cfg, precision="fp32", device="cpu")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m.attach_rankers()
    assert m.text_ranker is None and m.visual_ranker is None and len(w) == 2
    marker = object()
    m.text_ranker = marker
    m.attach_rankers()
    assert m.text_ranker is marker          # an attached ranker is never replaced


def test_last_text_hidden_state_semantics_do_not_depend_on_the_installed_transformers():
    """hidden_states[num_hidden_layers] is the last layer's output BEFORE final_norm in transformers
    4.48 - 4.5x (what the reference pins and the Judge was trained with) and the normalised tensor in 5.x; the reference's
    default nth_text_layer = 22 selects exactly that entry.  `last_text_layer_prenorm` (default True = 4.x) decides, on the
    checker's tower (tests/torch_text.py) as on the HIP one (tests/test_mbert_gpu.py)."""
    from tests.torch_text import TorchTextTower
    tm = G.make_text_model(SAMAudioJudgeConfig(text_model=G.TINY_TEXT), seed=5) if hasattr(G, "make_text_model") else None
    if tm is None:
        import transformers
        torch.manual_seed(5)
        tm = transformers.ModernBertModel(transformers.ModernBertConfig(**G.TINY_TEXT)).eval()
    L = tm.config.num_hidden_layers
    tower = TorchTextTower(tm)
    tower.place("cpu")
    ids = torch.randint(3, 128, (2, 7), generator=torch.Generator().manual_seed(1))
    mask = torch.ones(2, 7, dtype=torch.long)
    grabbed = []
    hook = tm.final_norm.register_forward_pre_hook(lambda mod, args: grabbed.append(args[0]))
    with torch.inference_mode():
        ref = tm(input_ids=ids, attention_mask=mask, output_hidden_states=True)
    hook.remove()
    with torch.inference_mode():
        pre = tower.hidden(ids, mask, L)                        # default: 4.x meaning
        post = tower.hidden(ids, mask, L, last_prenorm=False)   # 5.x meaning
        mid = tower.hidden(ids, mask, L - 1)
    assert torch.equal(pre, grabbed[0]) and torch.equal(post, ref.last_hidden_state)
    assert torch.equal(mid, ref.hidden_states[L - 1]) and not torch.allclose(pre, post)
    assert SAMAudioJudgeConfig().last_text_layer_prenorm is True
