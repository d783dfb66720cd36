"""Host logic and the C-ABI surface, no GPU needed: the library loads, exports every symbol that
include/samaudio.h declares, and its host-side validation behaves like the reference's error surface."""
import ctypes as C
import os
import re

import pytest
import torch

from sam_audio_amd import SAMAudioProcessor, hip, preset_config
from sam_audio_amd.config import SAMAudioConfig, TransformerConfig
from sam_audio_amd.synthetic import init_state_dict
from sam_audio_amd.weights import (_head_major, _interleave16, convert_codec, convert_dit, expected_keys,
                                   split_missing_unexpected)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "samaudio.h")).read()
    declared = set(re.findall(r"\b(samaudio_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for operands in ("bf16", "fp16"):          # the two builds of the library (16-bit operand format), same C ABI
        lib = hip.lib(operands)
        for name in sorted(declared):
            assert hasattr(lib, name), f"{hip.LIB_PATHS[operands]} does not export {name}"
        assert b"gfx950" in lib.samaudio_version()
        assert (b"fp16" in lib.samaudio_version()) == (operands == "fp16")
    assert declared == set(hip.EXPORTED_SYMBOLS)


def test_config_defaults_match_reference_rule():
    t = TransformerConfig()
    assert (t.dim, t.n_heads, t.n_layers, t.head_dim, t.ffn_hidden, t.rope_theta) == (2048, 16, 16, 128, 5504, 20000.0)
    assert preset_config("large*").transformer.ffn_hidden == 7552
    assert preset_config("small*").transformer.ffn_hidden == 4096
    assert SAMAudioConfig().audio_codec.hop_length == 1920
    SAMAudioConfig(transformer=dict(dim=2048, n_heads=32)).check_supported()   # head_dim 64: built since round 5
    with pytest.raises(NotImplementedError):
        SAMAudioConfig(transformer=dict(dim=2048, n_heads=64)).check_supported()   # head_dim 32 is not
    with pytest.raises(TypeError):
        SAMAudioConfig(transformer=dict(bogus=1))


def test_weight_relayout_is_a_permutation_of_reference_semantics():
    H, hd, K = 3, 128, 64
    w = torch.randn(H * hd, K)
    x = torch.randn(5, K)
    ref = (x @ w.T).reshape(5, hd, H).permute(0, 2, 1)             # reference reshape_heads: c = d*H + h
    mine = (x @ _head_major(w, H).T).reshape(5, H, hd)             # head-major columns
    assert torch.equal(ref, mine)
    w1, w3 = torch.randn(64, 8), torch.randn(64, 8)
    w13 = _interleave16(w1, w3)
    assert torch.equal(w13[0:16], w1[0:16]) and torch.equal(w13[16:32], w3[0:16]) and torch.equal(w13[32:48], w1[16:32])


def test_state_dict_key_bookkeeping():
    cfg = preset_config("tiny")
    keys = expected_keys(cfg)
    assert "transformer.layers.1.cross_attention.k_norm.weight" in keys and "audio_codec.decoder.model.6.weight" in keys
    missing, unexpected = split_missing_unexpected(keys + ["text_encoder.model.x", "bogus.weight"], cfg)
    assert missing == [] and unexpected == ["bogus.weight"]
    missing, _ = split_missing_unexpected([k for k in keys if k != "proj.bias"], cfg)
    assert missing == ["proj.bias"]
    wn = [k.replace(".weight", ".weight_g") if k.endswith("decoder.model.0.weight") else k for k in keys]
    wn.append("audio_codec.decoder.model.0.weight_v")
    assert split_missing_unexpected(wn, cfg) == ([], [])


def _ctx(cfg, precision):
    t, c = cfg.transformer, cfg.audio_codec
    hc = hip.Config(precision=precision, dim=t.dim, n_heads=t.n_heads, n_layers=t.n_layers, ffn_hidden=t.ffn_hidden,
                    latent_channels=256, text_dim=768, video_dim=1024, freq_dim=256, anchor_dim=128, anchor_vocab=4,
                    max_positions=t.max_positions, norm_eps=1e-5, codec_dim=128, codec_latent=1024, enc_dim=64,
                    dec_dim=1536, enc_rates=(C.c_int32 * 4)(2, 8, 10, 12), dec_rates=(C.c_int32 * 4)(12, 10, 8, 2))
    ctx = C.c_void_p()
    hip.check(hip.lib().samaudio_create(C.byref(hc), C.byref(ctx)))
    return ctx


@pytest.mark.parametrize("prec,dtype", [(hip.F32, torch.float32), (hip.BF16, torch.bfloat16)])
def test_engine_accepts_converted_weights_and_reports_missing_ones(prec, dtype):
    """set_tensor/finalize only look at names, dtypes and shapes, so CPU pointers are fine here."""
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=0)
    lib = hip.lib()
    ctx = _ctx(cfg, prec)
    keep = []
    for group, what in ((convert_dit(sd, cfg, dtype, "cpu"), 0), (convert_codec(sd, cfg, dtype, "cpu"), 1)):
        with pytest.raises(RuntimeError, match="missing weight"):
            hip.check(lib.samaudio_finalize(ctx, what))
        for name, t in group.items():
            keep.append(t)
            dt = hip.DT_BF16 if t.dtype == torch.bfloat16 else hip.DT_F32
            hip.check(lib.samaudio_set_tensor(ctx, name.encode(), hip.ptr(t), dt, t.dim(), hip.shape_array(t.shape)))
        hip.check(lib.samaudio_finalize(ctx, what))
    bad = torch.zeros(3, 3)
    hip.check(lib.samaudio_set_tensor(ctx, b"final_norm", hip.ptr(bad), hip.DT_F32, 2, hip.shape_array(bad.shape)))
    with pytest.raises(RuntimeError, match="final_norm"):
        hip.check(lib.samaudio_finalize(ctx, 0))
    small = lib.samaudio_workspace_bytes(ctx, 1, 250, 8, 0, 0)
    big = lib.samaudio_workspace_bytes(ctx, 8, 250, 8, 0, 0)
    codec = lib.samaudio_workspace_bytes(ctx, 0, 0, 0, 2, 480000)
    assert 0 < small < big and codec > 2 * 480000 * 64 * 4
    with pytest.raises(hip.SamAudioHipError):  # no workspace yet -> state/workspace error, not a crash
        hip.check(lib.samaudio_prepare(ctx, 1, 8, 1, C.c_void_p(1 << 20), None, None, None, None, 0, None, None, None))
    lib.samaudio_destroy(ctx)


def test_ode_options_are_validated():
    from sam_audio_amd.model import DFLT_ODE_OPT, ode_grid
    method, grid = ode_grid(DFLT_ODE_OPT)
    assert method == hip.ODE_MIDPOINT and len(grid) == 17 and grid[0] == 0.0 and grid[-1] == 1.0
    assert ode_grid({"method": "euler", "options": {"step_size": 0.3}})[1] == [0.0, 0.3, 0.6, 0.8999999999999999, 1.0]
    with pytest.raises(ValueError):
        ode_grid({"method": "dopri5", "options": {"step_size": 0.1}})
    with pytest.raises(ValueError):
        ode_grid({"method": "midpoint", "options": {"rtol": 1e-3}})


def test_processor_batching_matches_reference_semantics():
    proc = SAMAudioProcessor(1920, 48000)
    a, b = torch.randn(2, 4000), torch.randn(1, 1920 * 3)
    batch = proc(descriptions=["a", "b"], audios=[a, b])
    assert batch.audios.shape == (2, 1, 5760)
    assert torch.allclose(batch.audios[0, 0, :4000], a.mean(0)) and float(batch.audios[0, 0, 4000:].abs().max()) == 0
    assert batch.sizes.tolist() == [3.0, 3.0] and batch.wav_sizes.tolist() == [4000, 5760]
    assert batch.audio_pad_mask.all() and batch.anchor_ids.tolist() == [[0, 3], [0, 3]]
    short = proc(descriptions=["a", "b"], audios=[a[:, :1000], b])
    assert short.audio_pad_mask.tolist() == [[True, False, False], [True, True, True]]
    assert short.anchor_alignment.tolist() == [[0, 1, 1], [0, 0, 0]]
    with pytest.raises(AssertionError):
        proc(descriptions=["a"], audios=[a, b])
    with pytest.raises(FileNotFoundError):  # paths are decoded (PCM WAV, see test_wav_paths_...); a missing file is just that
        proc(descriptions=["a"], audios=["file.wav"])
    masked = proc.mask_videos([torch.ones(2, 3, 4, 4)], [torch.tensor([[[[1.0]]]]).expand(2, 3, 4, 4)])
    assert float(masked[0].abs().max()) == 0


def test_anchor_range_guarantee_follows_the_tensors():
    """Batch.process_anchors range-checks ids / alignment on the host (real checks) and records the vocabulary they were checked
    against; SAMAudio.separate() skips its blocking device-side check only while that guarantee holds: Batch.to() and
    dist.shard_batch keep it, assigning either tensor by hand drops it."""
    from sam_audio_amd.dist import shard_batch
    from sam_audio_amd.processor import ANCHOR_VOCAB
    cfg = preset_config("tiny")
    hop = cfg.audio_codec.hop_length
    proc = SAMAudioProcessor.from_config(cfg)
    clips = [torch.zeros(1, 8 * hop), torch.zeros(1, 6 * hop)]
    batch = proc(descriptions=["a", "b"], audios=clips, anchors=[[("+", 0.0, 0.08)], [("-", 0.04, 0.12)]])
    assert batch.anchors_validated and batch.anchor_vocab_validated == len(ANCHOR_VOCAB) == cfg.num_anchors + 1
    assert batch.to("cpu").anchors_validated                              # moved, same values: kept
    assert shard_batch(batch, 1, 2).anchors_validated                      # rows of checked tensors: kept
    batch.anchor_ids = batch.anchor_ids.clone()                           # rebound by hand: dropped
    assert not batch.anchors_validated and batch.anchor_vocab_validated == 0
    batch.process_anchors(None)                                           # rebuilt (and re-checked) on the host: back
    assert batch.anchors_validated
    batch.anchor_alignment = batch.anchor_alignment + 0
    assert not batch.anchors_validated
    with pytest.raises(KeyError):                                         # a token outside the vocabulary never becomes an id
        batch.process_anchors([[("?", 0.0, 0.04)], []])


def test_split_weight_forms_of_the_x3_precisions():
    """weights.x3_weight / convert_dit_x3 / convert_codec_x3 (include/samaudio.h SAMAUDIO_OPT_X3_CLASSES): [W_hi | W_lo | W_hi] along K
    per input-channel block, hi + lo = W to ~2^-22, K-tile-major storage a pure permutation; precision plumbing of hip.py."""
    from sam_audio_amd.weights import convert_codec_x3, convert_dit_x3, ktm_to_rows, x3_weight
    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 128, generator=g) * torch.logspace(-4, 1, 128)[None]
    rows = x3_weight(w, torch.float16, ktm=False)
    assert rows.shape == (96, 384) and rows.dtype == torch.float16
    hi, lo = rows[:, :128].float(), rows[:, 128:256].float()
    assert torch.equal(rows[:, :128], rows[:, 256:]) and torch.equal(rows[:, :128], w.half())
    assert ((hi + lo - w).abs() <= w.abs() * 2.0 ** -21 + 2.0 ** -25).all()
    assert torch.equal(ktm_to_rows(x3_weight(w, torch.float16, ktm=True)), rows)
    assert torch.isfinite(x3_weight(torch.tensor([[1e5, -7e4] + [0.0] * 62]), torch.float16, ktm=False).float()).all()   # clamped, not inf
    cfg = preset_config("tiny")
    sd = init_state_dict(cfg, seed=1)
    dit = convert_dit(sd, cfg, torch.float32, torch.device("cpu"))
    t = cfg.transformer
    x3 = convert_dit_x3(dit, t.n_layers, torch.float16, hip.CLS_X3_DEFAULT)
    D, F = t.dim, t.ffn_hidden
    assert x3["L0.wqkv.x3"].shape == (3 * D // 64, 3 * D, 64) and x3["L0.w2.x3"].shape == (3 * F // 64, D, 64)
    assert x3["c_wkv_all.x3"].shape == (3 * D // 64, t.n_layers * 2 * D, 64)
    p3 = ktm_to_rows(x3["patch1.w.x3"])            # [D, 3 taps x 3D]: every tap's D columns split on its own
    for tap in range(3):
        blk = p3[:, tap * 3 * D:(tap + 1) * 3 * D]
        assert torch.equal(blk[:, :D], dit["patch1.w"][:, tap * D:(tap + 1) * D].half()) and torch.equal(blk[:, :D], blk[:, 2 * D:])
    assert set(k.rsplit(".", 2)[-2] for k in x3 if k.startswith("L0.")) == set(hip.X3_WEIGHTS)
    assert not convert_dit_x3(dit, t.n_layers, torch.float16, hip.CLS["w2"]).keys() - {f"L{i}.w2.x3" for i in range(t.n_layers)}
    codec = convert_codec(sd, cfg, torch.float32, torch.device("cpu"))
    c3 = convert_codec_x3(codec, torch.float16)
    assert c3 and all(codec[k[:-3]].shape[0] >= 256 for k in c3)            # wide convolutions only
    for k, v in c3.items():                                                 # [N, K / Cin, 3 Cin], blocks of the fp32 weight
        n, kk = codec[k[:-3]].shape
        cin = v.shape[2] // 3
        assert v.shape == (n, kk // cin, 3 * cin) and torch.equal(v[:, :, :cin], codec[k[:-3]].reshape(n, kk // cin, cin).half())
    assert "dec.s1.r0.w1.x3" in c3 and c3["dec.s1.r0.w1.x3"].shape[2] == 3 * 384 and "dec.s3.r0.w1.x3" not in c3
    # the narrow convolutions: "<name>.fly" twins, the weight split once in the fragment layout of GemmParams.flags bit 14
    from sam_audio_amd.weights import convert_codec_fly16, fly16_to_f32, fly16_weight
    fly = convert_codec_fly16(codec, torch.float16)
    assert fly and not ({k[:-4] for k in fly} & {k[:-3] for k in c3}) and "dec.s3.r0.w1.fly" in fly
    assert {k[:-4] for k in fly} | {k[:-3] for k in c3} == {k for k, v in codec.items() if v.dim() == 2 and v.shape[1] % 32 == 0}
    for k, v in fly.items():
        wb = codec[k[:-4]]
        assert v.dtype == torch.float16 and v.shape == (wb.shape[0], 2 * wb.shape[1])
        assert ((fly16_to_f32(v) - wb).abs() <= wb.abs() * 2.0 ** -21 + 2.0 ** -25).all()
    one = fly16_weight(torch.arange(64, dtype=torch.float32)[None] + 1, torch.float16)   # slab 0 of row 0: chunk c = k in {4c.., 16+4c..}
    assert one[0, :8].tolist() == [1, 2, 3, 4, 17, 18, 19, 20] and one[0, 8:16].tolist() == [5, 6, 7, 8, 21, 22, 23, 24]
    assert one[0, 32:64].abs().max() == 0 and one[0, 64:72].tolist() == [33, 34, 35, 36, 49, 50, 51, 52]
    assert hip.precision_code("fp16x3") == hip.F32 and hip.operands_for("fp16x3") == "fp16" and hip.operands_for("bf16x3") == "bf16"
    assert hip.storage_precision("fp16x3") == "fp32" and hip.act_dtype("fp16x3") == torch.float32
    assert hip.class_mask("attn,qkv") == hip.X3_ATTENTION | hip.CLS["qkv"]
    with pytest.raises(ValueError):
        hip.check_precision("fp16x3")              # the towers beside the DiT have no compensated mode
    hip.check_precision("fp16x3", x3_ok=True)


def test_product_path_has_no_cpu_fallback():
    """The host class refuses to run without a GPU instead of silently computing elsewhere."""
    from sam_audio_amd import SAMAudio
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = SAMAudio(preset_config("tiny"), precision="fp32", device="cpu")
    with pytest.raises(hip.SamAudioHipError):
        model.load_state_dict(init_state_dict(preset_config("tiny"), 0))
    src = "".join(open(os.path.join(ROOT, "sam_audio_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "sam_audio_amd")) if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_streams_argument_is_validated():
    from sam_audio_amd import SAMAudio
    cfg = preset_config("tiny")
    for ok in (1, 2):
        assert SAMAudio(cfg, precision="bf16", streams=ok).streams == ok
    with pytest.raises(ValueError):
        SAMAudio(cfg, precision="bf16", streams=3)


def test_bench_core_detection_and_gemm_params_mirror():
    """bench.py's cpu_baseline must not oversubscribe a cgroup-limited box, and the ctypes mirror of GemmParams must
    have the size the library checks in samaudio_op_gemm (a stale mirror would only show up on the GPU otherwise)."""
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    p = hip.GemmParams()
    # an empty problem is rejected for its contents (ERR_ARG), a size mismatch would be rejected with the same code
    # but a different message
    rc = hip.lib().samaudio_op_gemm(C.byref(p), C.sizeof(p), hip.BF16, None)
    assert rc == hip.ERR_ARG and b"size mismatch" not in hip.lib().samaudio_last_error()
    rc = hip.lib().samaudio_op_gemm(C.byref(p), C.sizeof(p) - 8, hip.BF16, None)
    assert rc == hip.ERR_ARG and b"size mismatch" in hip.lib().samaudio_last_error()


def test_wav_paths_are_decoded_like_torchaudio_load(tmp_path):
    """PCM WAV at the model rate -> the tensor torchaudio.load would return (int16 / 32768 etc.); anything that would
    need torchaudio (other rates, compressed formats) raises instead of guessing."""
    import wave
    import numpy as np
    from sam_audio_amd.processor import batch_audio, load_wav
    rng = np.random.default_rng(0)
    pcm = rng.integers(-32768, 32767, size=(1000, 2), dtype=np.int16)
    p16 = str(tmp_path / "a.wav")
    with wave.open(p16, "wb") as f:
        f.setnchannels(2); f.setsampwidth(2); f.setframerate(48000); f.writeframes(pcm.tobytes())
    x = load_wav(p16, 48000)
    assert x.shape == (2, 1000) and x.dtype == torch.float32
    assert torch.equal(x, torch.from_numpy(pcm.T.astype(np.float32) / 32768.0))
    v24 = rng.integers(-(1 << 23), (1 << 23) - 1, size=500, dtype=np.int32)
    p24 = str(tmp_path / "b.wav")
    with wave.open(p24, "wb") as f:
        f.setnchannels(1); f.setsampwidth(3); f.setframerate(48000)
        f.writeframes(b"".join(int(s).to_bytes(3, "little", signed=True) for s in v24))
    assert torch.equal(load_wav(p24, 48000)[0], torch.from_numpy(v24.astype(np.float32) / 8388608.0))
    wavs, sizes = batch_audio([p16, torch.zeros(1, 1200)], 48000)
    assert wavs.shape == (2, 1, 1200) and sizes.tolist() == [1000, 1200]
    assert torch.allclose(wavs[0, 0, :1000], x.mean(0))
    # a file at another rate is resampled to the model's (reference processor.py:29-30): 1000 samples at 48 kHz read as
    # if the model ran at 44.1 kHz -> ceil(1000 * 441 / 480) samples
    assert load_wav(p16, 44100).shape == (2, 919)
    bad = tmp_path / "c.wav"
    bad.write_bytes(b"not a wav file")
    with pytest.raises(ValueError, match="PCM WAV"):
        load_wav(str(bad), 48000)


def test_sinc_resampler_properties():
    """`processor.resample` (restated torchaudio.functional.resample, UNPINNED offline): output length
    ceil(new * n / orig); identity at equal rates; a band-limited tone keeps frequency, phase and amplitude; DC gain 1;
    content above the new Nyquist rate is removed when down-sampling; agrees with scipy's polyphase resampler on a
    band-limited signal (different filter design, so only to ~1e-2)."""
    import math
    import numpy as np
    import scipy.signal
    from sam_audio_amd.processor import resample
    x = torch.randn(2, 3, 4410)
    assert resample(x, 44100, 44100) is x
    for o, n in ((44100, 48000), (16000, 48000), (96000, 48000), (22050, 48000)):
        y = resample(x, o, n)
        assert y.shape == (2, 3, math.ceil(n * 4410 / o)) and y.dtype == x.dtype
    t44 = torch.arange(44100, dtype=torch.float64) / 44100
    tone = torch.sin(2 * math.pi * 1000 * t44 + 0.3).float()
    y = resample(tone, 44100, 48000)
    t48 = torch.arange(y.numel(), dtype=torch.float64) / 48000
    want = torch.sin(2 * math.pi * 1000 * t48 + 0.3).float()
    assert (y - want)[200:-200].abs().max().item() < 2e-3
    assert (resample(torch.ones(1, 8000), 16000, 48000)[0, 100:-100] - 1).abs().max().item() < 2e-3
    hi = torch.sin(2 * math.pi * 30000 * torch.arange(96000, dtype=torch.float64) / 96000).float()   # above 24 kHz
    assert resample(hi, 96000, 48000)[200:-200].abs().max().item() < 2e-2
    g = torch.Generator().manual_seed(0)
    sig = sum(torch.sin(2 * math.pi * f * t44 + p) for f, p in ((220.0, 0.1), (1300.0, 1.0), (5200.0, 2.0))).float() / 3
    ours = resample(sig, 44100, 48000).numpy()
    ref = scipy.signal.resample_poly(sig.numpy().astype(np.float64), 160, 147)
    assert np.abs(ours - ref[: ours.size])[500:-500].max() < 1e-2
