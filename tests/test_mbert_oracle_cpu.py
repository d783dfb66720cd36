"""oracle/mbert_oracle.py (the CPU restatement the ModernBERT text tower on the HIP library is checked against, SURVEY.md
section 8 rows a17 / a18) pinned to `transformers.ModernBertModel` - the reference's own dependency
(reference sam_audio/model/judge.py:48,74-88) - and the host-side re-layout of sam_audio_amd/mbert_encoder.py."""
import pytest
import torch
import transformers

from oracle import mbert_oracle as O
from sam_audio_amd import mbert_encoder as E

SMALL = dict(hidden_size=64, intermediate_size=96, num_hidden_layers=4, num_attention_heads=2, vocab_size=128, pad_token_id=0,
             bos_token_id=1, eos_token_id=2, cls_token_id=1, sep_token_id=2, global_attn_every_n_layers=3, local_attention=8,
             max_position_embeddings=256)


def _model(seed, **kw):
    cfg = transformers.ModernBertConfig(**kw)
    cfg._attn_implementation = "eager"
    torch.manual_seed(seed)
    m = transformers.ModernBertModel(cfg).eval()
    with torch.no_grad():   # default init keeps the norms at 1: make every weight count
        for n, p in m.named_parameters():
            if "norm" in n:
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            else:
                p.mul_(3.0)
    return m, cfg


def _inputs(cfg, B, L, seed):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(3, cfg.vocab_size, (B, L), generator=g)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    lens[0] = L
    mask = (torch.arange(L)[None, :] < lens[:, None]).long()
    return ids, mask


@pytest.mark.parametrize("kw,B,L", [(SMALL, 3, 40), (dict(SMALL, global_attn_every_n_layers=2, num_hidden_layers=3), 2, 9),
                                    (dict(hidden_size=768, intermediate_size=1152, num_hidden_layers=4, num_attention_heads=12,
                                          vocab_size=50368), 2, 150)])
def test_oracle_matches_transformers_modernbert(kw, B, L):
    m, cfg = _model(3, **kw)
    ids, mask = _inputs(cfg, B, L, 4)
    dims = E.MBertDims.from_hf(cfg)
    with torch.inference_mode():
        ref = m(input_ids=ids, attention_mask=mask, output_hidden_states=True)
        states, last = O.mbert_hidden_states(m.state_dict(), dims, ids, mask)
    assert len(ref.hidden_states) == cfg.num_hidden_layers + 1 == len(states)
    valid = mask.bool()
    for n, (got, want) in enumerate(zip(states, ref.hidden_states)):
        err = (got - want)[valid].abs().max().item() / want[valid].abs().max().item()
        assert err < 5e-6, f"hidden_states[{n}]: {err}"
    err = (last - ref.last_hidden_state)[valid].abs().max().item() / ref.last_hidden_state[valid].abs().max().item()
    print(f"mbert oracle vs transformers: last_hidden_state rel {err:.2e}")
    assert err < 5e-6
    # the other generation's hidden_states[layers] (transformers 4.48 - 4.5x: before final_norm) = what goes INTO the
    # installed module's final_norm, whatever that module then records as its last hidden state
    grabbed = []
    hook = m.final_norm.register_forward_pre_hook(lambda mod, args: grabbed.append(args[0]))
    with torch.inference_mode():
        m(input_ids=ids, attention_mask=mask)
        pre, last4 = O.mbert_hidden_states(m.state_dict(), dims, ids, mask, last_prenorm=True)
    hook.remove()
    err = (pre[-1] - grabbed[0])[valid].abs().max().item() / grabbed[0][valid].abs().max().item()
    assert err < 5e-6 and torch.equal(last4, last)
    assert not torch.allclose(pre[-1], states[-1]) and all(torch.equal(a, b) for a, b in zip(pre[:-1], states[:-1]))


def test_host_relayout_keys_and_rope_tables():
    m, cfg = _model(5, **SMALL)
    dims = E.MBertDims.from_hf(cfg, max_len=64)
    assert dims.global_attn_every_n_layers == 3 and dims.local_attention == 8 and dims.max_len == 64
    assert dims.global_rope_theta == 160000.0 and dims.local_rope_theta == 10000.0
    assert set(E.expected_keys(dims)) == set(m.state_dict())
    cos, sin = E.rope_tables(dims)
    assert cos.shape == (2, 64, 32)
    pos = torch.arange(64)[None]
    for j, kind in enumerate(("full_attention", "sliding_attention")):
        c, s = m.rotary_emb(torch.zeros(1, 64, 64), pos, kind)
        assert torch.equal(cos[j], c[0]) and torch.equal(sin[j], s[0])
    conv = E.convert_mbert(m.state_dict(), dims, torch.float32, "cpu")
    assert "L0.ln1" not in conv and conv["L1.ln1"].shape == (64,) and conv["L2.wi"].shape == (192, 64)
    with pytest.raises(NotImplementedError):
        E.MBertDims.from_hf(transformers.ModernBertConfig(**dict(SMALL, norm_bias=True))).check_supported()
