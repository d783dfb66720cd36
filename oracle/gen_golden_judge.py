"""Pin oracle/judge_oracle.py and mint tests/golden/judge_*.npz.   Run in the build container:

    python -m oracle.gen_golden_judge

1. `peav_transformer` against Hugging Face transformers' `PeAudioEncoder` (the in-container port of the un-vendored
   perception_models PE-AV transformer): the HF module is instantiated with seeded weights, its DAC embedder is
   bypassed (the Judge feeds already-projected features, reference judge.py:108-111) and its own forward() runs the
   patch embedder, layers, final norm and output projection.
2. `judge_forward` against the reference's OWN `SAMAudioJudgeModel.forward` (sam_audio/model/judge.py:90-132,
   imported read-only through oracle/ref_import.py).  The two un-vendored classes it instantiates are bound to
   adapters: `PEAVTransformer` -> the HF encoder of step 1, `dacvae.DACVAE` -> this oracle's DAC encoder restatement;
   the ModernBERT text tower is a seeded random-init `transformers.ModernBertModel` (tiny config).
3. Integer fixture: spans -> `Batch.process_anchors` round trip of `spans_from_logits` (reference processor.py:78-124).

The fixtures store the REFERENCE-side outputs; tests/test_judge_oracle.py re-checks the restatement against them on
every run (no /root/reference needed at test time; step 1 is re-run live when transformers ships pe_audio).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import judge_oracle as J  # noqa: E402
from oracle import samaudio_oracle as O  # noqa: E402
from sam_audio_amd.config import PEAVTransformerConfig, SAMAudioJudgeConfig  # noqa: E402
from sam_audio_amd.synthetic import init_judge_state_dict, init_peav_state_dict  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

TINY_TC = dict(hidden_size=256, intermediate_size=448, num_hidden_layers=2, num_attention_heads=2)
TINY_FT = dict(hidden_size=256, intermediate_size=320, num_hidden_layers=1, num_attention_heads=2)
TINY_TEXT = dict(hidden_size=64, intermediate_size=96, num_hidden_layers=3, num_attention_heads=2, vocab_size=128,
                 pad_token_id=0, bos_token_id=1, eos_token_id=2, cls_token_id=1, sep_token_id=2,
                 global_attn_every_n_layers=2, local_attention=8, max_position_embeddings=64)
TINY_CODEC = dict(encoder_dim=8, encoder_rates=[2, 2, 4, 4], latent_dim=64, decoder_dim=64,
                  decoder_rates=[4, 4, 2, 2], codebook_dim=64)


def tiny_judge_config() -> SAMAudioJudgeConfig:
    return SAMAudioJudgeConfig(audio_codec=TINY_CODEC, transformer=TINY_TC, finetune_transformer=TINY_FT,
                               text_model=TINY_TEXT, nth_text_layer=2, bottleneck_dim=64)


def hf_encoder(tc: PEAVTransformerConfig, sd, prefix):
    """HF PeAudioEncoder with `sd[prefix + ...]` loaded and the DAC embedder bypassed."""
    from transformers.models.pe_audio.configuration_pe_audio import PeAudioEncoderConfig
    from transformers.models.pe_audio.modeling_pe_audio import PeAudioEncoder

    hc = PeAudioEncoderConfig(
        hidden_size=tc.hidden_size, intermediate_size=tc.intermediate_size, num_hidden_layers=tc.num_hidden_layers,
        num_attention_heads=tc.num_attention_heads, head_dim=tc.head_dim, rms_norm_eps=tc.rms_norm_eps,
        rope_parameters={"rope_theta": tc.rope_theta, "rope_type": "default"}, attention_bias=tc.attention_bias,
        dac_config=dict(encoder_hidden_size=4, downsampling_ratios=[2, 2], codebook_dim=8, hidden_size=16))
    hc._attn_implementation = "eager"
    m = PeAudioEncoder(hc).eval()
    own = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    missing, unexpected = m.load_state_dict(own, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("embedder.") for k in missing), missing

    class Bypass(torch.nn.Module):
        def forward(self, input_values, padding_mask=None):
            return input_values, padding_mask

    m.embedder = Bypass()
    return m


def pin_transformer():
    tc = PEAVTransformerConfig(**TINY_TC)
    g = torch.Generator().manual_seed(5)
    sd = init_peav_state_dict(tc, "t.", g, torch.device("cpu"))
    B, T = 3, 21
    x = torch.randn(B, T, tc.hidden_size, generator=g)
    sizes = torch.tensor([21, 13, 6])
    mask = torch.arange(T)[None] < sizes[:, None]
    m = hf_encoder(tc, sd, "t.")
    out = {}
    with torch.inference_mode():
        for name, pm in (("masked", mask), ("nomask", None)):
            ref = m(input_values=x, padding_mask=pm)
            last, pooled = J.peav_transformer(sd, "t.", x, pm, n_heads=tc.num_attention_heads,
                                              n_layers=tc.num_hidden_layers, eps=tc.rms_norm_eps,
                                              rope_theta=tc.rope_theta)
            valid = mask if pm is not None else torch.ones_like(mask)
            e1 = ((ref.last_hidden_state - last).abs() * valid[..., None]).max().item()
            e2 = (ref.pooler_output - pooled).abs().max().item()
            print(f"peav_transformer[{name}] vs HF PeAudioEncoder: last {e1:.2e} pooled {e2:.2e}")
            assert e1 < 2e-5 and e2 < 2e-5
            out[f"{name}_last"] = ref.last_hidden_state.numpy()
            out[f"{name}_pooled"] = ref.pooler_output.numpy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "judge_peav_tiny.npz"), **out)


def judge_inputs(cfg):
    g = torch.Generator().manual_seed(21)
    B = 2
    hop = cfg.audio_codec.hop_length
    T = 12
    lengths = torch.tensor([T * hop, 7 * hop])
    wav_in = 0.3 * torch.randn(B, 1, T * hop, generator=g)
    wav_sep = 0.3 * torch.randn(B, 1, T * hop, generator=g)
    pad = torch.arange(T * hop)[None] < lengths[:, None]
    wav_in = wav_in * pad[:, None]
    wav_sep = wav_sep * pad[:, None]
    ids = torch.randint(3, 128, (B, 6), generator=g)
    ids[:, 0] = 1
    att = torch.ones(B, 6, dtype=torch.long)
    att[1, 4:] = 0
    return dict(input_ids=ids, attention_mask=att, input_values=wav_in, separated_values=wav_sep, padding_mask=pad)


def text_tower(cfg, seed=33):
    import transformers
    torch.manual_seed(seed)
    tm = transformers.ModernBertModel(transformers.ModernBertConfig(**cfg.text_model)).eval()
    return tm


def text_pooled(tm, cfg, input_ids, attention_mask):
    """reference judge.py:76-88."""
    with torch.inference_mode():
        out = tm(input_ids=input_ids, attention_mask=attention_mask, output_hidden_states=cfg.nth_text_layer is not None)
        hs = out.last_hidden_state if cfg.nth_text_layer is None else out.hidden_states[cfg.nth_text_layer]
    return hs[:, 0]


def pin_judge():
    from oracle import ref_import
    ref_import.import_reference()
    import transformers
    from transformers.modeling_outputs import BaseModelOutputWithPooling
    import importlib
    RJ = importlib.import_module("sam_audio.model.judge")
    RC = importlib.import_module("sam_audio.model.codec")

    cfg = tiny_judge_config()
    sd = init_judge_state_dict(cfg, seed=9)
    inp = judge_inputs(cfg)
    tm = text_tower(cfg)

    class PEAVAdapter(torch.nn.Module):  # stands in for core.audio_visual_encoder.transformer.Transformer
        def __init__(self, tc):
            super().__init__()
            self.tc = tc
            self.inner = None

        def forward(self, x, padding_mask=None):
            if padding_mask is not None and padding_mask.shape[0] != x.shape[0]:
                padding_mask = padding_mask.repeat(x.shape[0] // padding_mask.shape[0], 1)  # see judge_oracle.py
            o = self.inner(input_values=x, padding_mask=padding_mask)
            return BaseModelOutputWithPooling(last_hidden_state=o.last_hidden_state, pooler_output=o.pooler_output)

    class DacAdapter(torch.nn.Module):  # stands in for dacvae.DACVAE: .encoder / .quantizer.in_proj
        def __init__(self, **kw):
            super().__init__()
            outer = self

            class Enc(torch.nn.Module):
                def forward(self, wav):
                    return outer._latent(wav)

            class Quant(torch.nn.Module):
                def in_proj(self, z):
                    return z

            self.encoder, self.quantizer = Enc(), Quant()

        def _latent(self, wav):
            # oracle DAC encoder, both halves of quantizer.in_proj (the reference keeps chunk(2)[0], codec.py:67-68)
            mean = O.dac_encode(sd, cfg.audio_codec, wav, prefix="audio_codec.")
            return torch.cat([mean, torch.zeros_like(mean)], dim=1)

    RJ.PEAVTransformer = PEAVAdapter
    RJ.BaseModelOutputWithPooling = BaseModelOutputWithPooling
    RC.dacvae.DACVAE = DacAdapter
    RJ.AutoModel = type("AutoModel", (), {"from_config": staticmethod(lambda c: tm)})

    ref_cfg = type("Cfg", (), {})()
    ref_cfg.audio_codec = type("C", (), dict(vars(cfg.audio_codec), hop_length=cfg.audio_codec.hop_length))()
    ref_cfg.transformer = cfg.transformer
    ref_cfg.finetune_transformer = cfg.finetune_transformer
    ref_cfg.text_model = type("T", (), {"hidden_size": cfg.text_hidden})()
    ref_cfg.nth_text_layer = cfg.nth_text_layer
    ref_cfg.bottleneck_dim = cfg.bottleneck_dim
    model = RJ.SAMAudioJudgeModel(ref_cfg).eval()
    model.transformer.inner = hf_encoder(cfg.transformer, sd, "transformer.")
    model.finetune_transformer.inner = hf_encoder(cfg.finetune_transformer, sd, "finetune_transformer.")
    own = {k: v for k, v in sd.items() if not k.startswith(("transformer.", "finetune_transformer.", "audio_codec."))}
    missing, unexpected = model.load_state_dict(own, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith(("transformer.", "finetune_transformer.", "text_model.", "audio_codec.")) for k in missing), missing
    with torch.inference_mode():
        res = model(**inp)
        ref = torch.cat([res.overall, res.recall, res.precision, res.faithfulness], dim=1)
        pooled = text_pooled(tm, cfg, inp["input_ids"], inp["attention_mask"])
        mine = J.judge_forward(sd, cfg, pooled, inp["input_values"], inp["separated_values"], inp["padding_mask"])
    err = (ref - mine).abs().max().item()
    print(f"judge_forward vs reference SAMAudioJudgeModel.forward: {err:.2e}\n  ref {ref.tolist()}")
    assert err < 5e-5
    np.savez_compressed(os.path.join(GOLDEN_DIR, "judge_tiny.npz"), scores=ref.numpy(), text_pooled=pooled.numpy())


def mint_spans():
    hop, sr = 1920, 48000
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(4, 60, generator=g) * 2
    sizes = torch.tensor([60, 60, 41, 17])
    pad = torch.arange(60)[None] < sizes[:, None]
    spans = J.spans_from_logits(logits, pad, hop, sr)
    anchors = [[("+", s, e) for s, e in row] for row in spans]
    ids, align = O.anchors_to_ids(anchors, pad, hop, sr)       # pinned restatement of Batch.process_anchors
    active = ((logits > 0) & pad)
    want = torch.where(active, torch.ones_like(align), torch.zeros_like(align))
    got = (align >= 2).long()
    assert torch.equal(got, want), "spans do not map back onto the active frames"
    np.savez_compressed(os.path.join(GOLDEN_DIR, "judge_spans.npz"), logits=logits.numpy(), sizes=sizes.numpy(),
                        ids=ids.numpy(), align=align.numpy())
    print("span round trip ok:", [len(r) for r in spans], "spans per row")


if __name__ == "__main__":
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    pin_transformer()
    pin_judge()
    mint_spans()
