"""oracle/judge_oracle.py against its committed pins (tests/golden/judge_*.npz, minted by oracle/gen_golden_judge.py from
the Hugging Face PeAudioEncoder and from the reference's own SAMAudioJudgeModel.forward), plus a live re-check against
the HF module (transformers travels with the image; /root/reference does not and is not needed here)."""
import os

import numpy as np
import pytest
import torch

from oracle import gen_golden_judge as G
from oracle import judge_oracle as J
from oracle import samaudio_oracle as O
from sam_audio_amd.config import PEAVTransformerConfig
from sam_audio_amd.synthetic import init_judge_state_dict, init_peav_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _peav_case():
    tc = PEAVTransformerConfig(**G.TINY_TC)
    g = torch.Generator().manual_seed(5)
    sd = init_peav_state_dict(tc, "t.", g, torch.device("cpu"))
    x = torch.randn(3, 21, tc.hidden_size, generator=g)
    mask = torch.arange(21)[None] < torch.tensor([21, 13, 6])[:, None]
    return tc, sd, x, mask


def test_peav_transformer_matches_hf_fixture():
    tc, sd, x, mask = _peav_case()
    gold = np.load(os.path.join(GOLDEN, "judge_peav_tiny.npz"))
    for name, pm in (("masked", mask), ("nomask", None)):
        last, pooled = J.peav_transformer(sd, "t.", x, pm, n_heads=tc.num_attention_heads,
                                          n_layers=tc.num_hidden_layers, eps=tc.rms_norm_eps, rope_theta=tc.rope_theta)
        valid = (mask if pm is not None else torch.ones_like(mask))[..., None]
        assert ((torch.from_numpy(gold[f"{name}_last"]) - last).abs() * valid).max() < 2e-5
        assert (torch.from_numpy(gold[f"{name}_pooled"]) - pooled).abs().max() < 2e-5


def test_peav_transformer_matches_hf_module_live():
    pytest.importorskip("transformers.models.pe_audio.modeling_pe_audio")
    tc, sd, x, mask = _peav_case()
    m = G.hf_encoder(tc, sd, "t.")
    with torch.inference_mode():
        ref = m(input_values=x, padding_mask=mask)
        last, pooled = J.peav_transformer(sd, "t.", x, mask, n_heads=tc.num_attention_heads,
                                          n_layers=tc.num_hidden_layers, eps=tc.rms_norm_eps, rope_theta=tc.rope_theta)
    assert ((ref.last_hidden_state - last).abs() * mask[..., None]).max() < 2e-5
    assert (ref.pooler_output - pooled).abs().max() < 2e-5


def test_judge_forward_matches_reference_fixture():
    cfg = G.tiny_judge_config()
    sd = init_judge_state_dict(cfg, seed=9)
    inp = G.judge_inputs(cfg)
    gold = np.load(os.path.join(GOLDEN, "judge_tiny.npz"))
    with torch.inference_mode():
        got = J.judge_forward(sd, cfg, torch.from_numpy(gold["text_pooled"]), inp["input_values"],
                              inp["separated_values"], inp["padding_mask"])
    assert (got - torch.from_numpy(gold["scores"])).abs().max() < 5e-5


def test_judge_rows_are_independent_and_input_dedup_is_exact():
    """The reference repeats the mixture once per candidate (ranking/judge.py:31-33); every op of the Judge is
    per-row, so scoring rows separately (what the HIP path's de-duplicated input branch relies on) changes nothing."""
    cfg = G.tiny_judge_config()
    sd = init_judge_state_dict(cfg, seed=9)
    inp = G.judge_inputs(cfg)
    gold = np.load(os.path.join(GOLDEN, "judge_tiny.npz"))
    tp = torch.from_numpy(gold["text_pooled"])
    with torch.inference_mode():
        both = J.judge_forward(sd, cfg, tp, inp["input_values"], inp["separated_values"], inp["padding_mask"])
        for b in range(2):
            one = J.judge_forward(sd, cfg, tp[b:b + 1], inp["input_values"][b:b + 1], inp["separated_values"][b:b + 1],
                                  inp["padding_mask"][b:b + 1])
            assert (one - both[b:b + 1]).abs().max() < 1e-5


def test_span_rule_round_trips_through_process_anchors_bit_exactly():
    gold = np.load(os.path.join(GOLDEN, "judge_spans.npz"))
    logits, sizes = torch.from_numpy(gold["logits"]), torch.from_numpy(gold["sizes"])
    pad = torch.arange(logits.shape[1])[None] < sizes[:, None]
    spans = J.spans_from_logits(logits, pad, 1920, 48000)
    ids, align = O.anchors_to_ids([[("+", s, e) for s, e in row] for row in spans], pad, 1920, 48000)
    assert np.array_equal(ids.numpy(), gold["ids"]) and np.array_equal(align.numpy(), gold["align"])
    assert torch.equal(align >= 2, (logits > 0) & pad)
    # the product's Batch runs the same integer pipeline (pinned to the reference's Batch by tests/golden/anchors.npz)
    from sam_audio_amd.processor import Batch
    b = Batch(audios=torch.zeros(4, 1, 60 * 1920), sizes=sizes, wav_sizes=sizes * 1920, descriptions=["x"] * 4,
              hop_length=1920, audio_sampling_rate=48000, audio_pad_mask=pad,
              anchors=[[("+", s, e) for s, e in row] for row in spans])
    assert torch.equal(b.anchor_alignment, align) and torch.equal(b.anchor_ids, ids)


def test_rerank_select_is_argmax_over_candidates():
    scores = torch.tensor([[0.1, 0.7, 0.3], [0.9, 0.2, 0.95]])
    assert J.rerank_select(scores).tolist() == [1, 2]
