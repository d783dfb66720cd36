"""oracle/t5_oracle.py (the CPU restatement the T5 prompt encoder on the HIP library is checked against, SURVEY.md
section 8 rows a3 / f4) pinned to `transformers.T5EncoderModel` - the reference's own dependency
(reference sam_audio/model/text_encoder.py:11-17) - and the host-side re-layout of sam_audio_amd/t5_encoder.py."""
import pytest
import torch
import transformers

from oracle import t5_oracle as T
from sam_audio_amd import t5_encoder as E


def _model(seed, **kw):
    cfg = transformers.T5Config(**kw)
    torch.manual_seed(seed)
    m = transformers.T5EncoderModel(cfg).eval()
    with torch.no_grad():   # T5's default init leaves the relative bias / norms near-trivial: make every term count
        for n, p in m.named_parameters():
            if "layer_norm" in n:
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif "relative_attention_bias" in n:
                p.copy_(torch.randn_like(p))
    return m, cfg


def _inputs(cfg, B, L, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, cfg.vocab_size, (B, L), generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    return ids, mask


@pytest.mark.parametrize("kw,B,L", [
    (dict(vocab_size=32128, d_model=768, d_kv=64, d_ff=3072, num_layers=2, num_heads=12, feed_forward_proj="relu"), 3, 9),
    (dict(vocab_size=100, d_model=64, d_kv=32, d_ff=128, num_layers=3, num_heads=2, feed_forward_proj="relu"), 4, 40),
    (dict(vocab_size=100, d_model=64, d_kv=16, d_ff=128, num_layers=2, num_heads=4, feed_forward_proj="gelu_new"), 2, 150),
])
def test_oracle_matches_transformers_t5_encoder(kw, B, L):
    m, cfg = _model(3, **kw)
    ids, mask = _inputs(cfg, B, L, 4)
    with torch.inference_mode():
        want = m(input_ids=ids, attention_mask=mask)["last_hidden_state"]
        got = T.t5_encoder(m.state_dict(), cfg, ids, mask)
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f"t5 oracle vs transformers: rel {err:.2e}")
    assert err < 2e-6


def test_bias_table_and_key_mapping_of_the_host_relayout():
    """The per-distance bias table equals what T5Attention.compute_bias gathers (all 1023 signed distances, incl. the
    float-edge buckets at 16 / 32 / 64), and the expected keys are exactly T5EncoderModel's (minus the tied copy)."""
    m, cfg = _model(5, vocab_size=50, d_model=64, d_kv=32, d_ff=128, num_layers=2, num_heads=2)
    dims = E.T5Dims.from_hf(cfg)
    assert dims.dense_act_fn == "relu" and not dims.is_gated_act and dims.max_len == 512
    att = m.encoder.block[0].layer[0].SelfAttention
    with torch.inference_mode():
        want = att.compute_bias(512, 512)[0]                     # [H, q, k]
    table = E.relative_bias_table(att.relative_attention_bias.weight, dims)
    q = torch.arange(512)
    got = table[:, (q[None, :] - q[:, None]) + 511]              # [H, q, k]
    assert torch.equal(got, want)
    keys = set(m.state_dict()) - {"encoder.embed_tokens.weight"}
    assert set(E.expected_keys(dims)) == keys
    gated = E.T5Dims.from_hf(transformers.T5Config(feed_forward_proj="gated-gelu"))
    assert gated.is_gated_act and gated.dense_act_fn == "gelu_new"
    with pytest.raises(NotImplementedError, match="gated"):
        gated.check_supported()
    conv = E.convert_t5(m.state_dict(), dims, torch.bfloat16, "cpu")
    assert conv["L1.wqkv"].shape == (3 * 64, 64) and conv["L1.wqkv"].dtype == torch.bfloat16
    assert conv["emb"].dtype == torch.float32 and conv["rel_bias"].shape == (2, 1023)
