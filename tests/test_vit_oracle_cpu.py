"""Pins oracle/vit_oracle.py (the restated PE-Core vision tower) against the
in-container implementations of the blocks it shares with other networks:

* everything but RoPE-2D and the attention pooling head == Hugging Face `CLIPVisionModelWithProjection` (strict
  key-for-key weight mapping, seeded weights);
* the attention pooling head == torch's own nn.MultiheadAttention / nn.LayerNorm / nn.Linear modules;
* RoPE-2D: structural properties of the published description (class token unrotated, norm preserving, x / y
  frequencies on the two halves of a head) - PARITY UNPINNED against perception_models (absent).
"""
import dataclasses

import pytest
import torch

from oracle import vit_oracle as V
from sam_audio_amd.config import PE_VISION_CONFIGS, PEVisionConfig
from sam_audio_amd.synthetic import init_vision_state_dict


def _clip_to_pe_keys(hf_sd, layers):
    sd = {
        "conv1.weight": hf_sd["vision_model.embeddings.patch_embedding.weight"],
        "class_embedding": hf_sd["vision_model.embeddings.class_embedding"],
        "positional_embedding": hf_sd["vision_model.embeddings.position_embedding.weight"],
        "ln_pre.weight": hf_sd["vision_model.pre_layrnorm.weight"],
        "ln_pre.bias": hf_sd["vision_model.pre_layrnorm.bias"],
        "ln_post.weight": hf_sd["vision_model.post_layernorm.weight"],
        "ln_post.bias": hf_sd["vision_model.post_layernorm.bias"],
        "proj": hf_sd["visual_projection.weight"].t().contiguous(),
    }
    used = 8
    for i in range(layers):
        s, d = f"vision_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        sd[d + "attn.in_proj_weight"] = torch.cat([hf_sd[s + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        sd[d + "attn.in_proj_bias"] = torch.cat([hf_sd[s + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        for a, b_ in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"),
                      ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
            sd[d + b_ + ".weight"], sd[d + b_ + ".bias"] = hf_sd[s + a + ".weight"], hf_sd[s + a + ".bias"]
        used += 16
    assert used == len(hf_sd), (used, len(hf_sd))   # strict: every HF tensor is consumed
    return sd


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_shared_blocks_match_hf_clip(act):
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    hcfg = CLIPVisionConfig(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2,
                            image_size=56, patch_size=14, projection_dim=64, hidden_act=act)
    hf = CLIPVisionModelWithProjection(hcfg).eval()
    with torch.no_grad():
        for p in hf.parameters():   # HF's default init leaves most tensors near zero: make every term count
            p.copy_(torch.randn_like(p) * (0.3 if p.ndim == 1 else p.shape[-1] ** -0.5))
    cfg = PEVisionConfig(image_size=56, patch_size=14, width=128, layers=3, heads=2, output_dim=64, mlp_ratio=4.0,
                         use_rope2d=False, pool_type="tok", act=act)
    sd = _clip_to_pe_keys(dict(hf.state_dict()), 3)
    x = torch.randn(3, 3, 56, 56)
    with torch.no_grad():
        want = hf(pixel_values=x).image_embeds
        got = V.vision_tower(sd, cfg, x)
    assert (got - want).abs().max().item() < 2e-5 * want.abs().max().item() + 1e-5


def test_attention_pool_matches_torch_modules():
    torch.manual_seed(1)
    cfg = PE_VISION_CONFIGS["pe-mini"]
    W, Fw = cfg.width, cfg.mlp_width
    attn = torch.nn.MultiheadAttention(W, cfg.attn_pooler_heads, batch_first=True).eval()
    ln = torch.nn.LayerNorm(W, eps=cfg.ln_eps)
    fc, pj = torch.nn.Linear(W, Fw), torch.nn.Linear(Fw, W)
    probe = torch.randn(1, 1, W)
    with torch.no_grad():
        for m in (attn, ln, fc, pj):
            for p in m.parameters():
                p.copy_(torch.randn_like(p) * (0.3 if p.ndim == 1 else p.shape[-1] ** -0.5))
    sd = {"attn_pool.probe": probe, "attn_pool.attn.in_proj_weight": attn.in_proj_weight,
          "attn_pool.attn.in_proj_bias": attn.in_proj_bias, "attn_pool.attn.out_proj.weight": attn.out_proj.weight,
          "attn_pool.attn.out_proj.bias": attn.out_proj.bias, "attn_pool.layernorm.weight": ln.weight,
          "attn_pool.layernorm.bias": ln.bias, "attn_pool.mlp.c_fc.weight": fc.weight, "attn_pool.mlp.c_fc.bias": fc.bias,
          "attn_pool.mlp.c_proj.weight": pj.weight, "attn_pool.mlp.c_proj.bias": pj.bias}
    x = torch.randn(4, cfg.tokens, W)
    with torch.no_grad():
        q = probe.repeat(4, 1, 1)
        y = attn(q, x, x, need_weights=False)[0]
        want = (y + pj(torch.nn.functional.gelu(fc(ln(y)))))[:, 0]
        got = V.attn_pool(sd, cfg, x)
    assert (got - want).abs().max().item() < 1e-5


def test_rope2d_structure():
    cfg = PE_VISION_CONFIGS["PE-Core-L14-336"]
    cos, sin = V.rope2d_tables(cfg, cfg.grid, cfg.grid)
    hd = cfg.width // cfg.heads
    assert cos.shape == (cfg.tokens, hd)
    assert torch.all(cos[0] == 1) and torch.all(sin[0] == 0)            # class token at (0, 0): identity
    ang = torch.atan2(sin, cos)
    assert torch.equal(ang[:, 0::2], ang[:, 1::2])                      # adjacent pairs share an angle
    g = cfg.grid
    tok = lambda y, x: 1 + y * g + x
    # first half of the head depends on x only, second half on y only; coordinates start at 1
    assert torch.allclose(cos[tok(3, 5), : hd // 2], cos[tok(7, 5), : hd // 2])
    assert torch.allclose(cos[tok(3, 5), hd // 2:], cos[tok(3, 9), hd // 2:])
    assert abs(ang[tok(0, 0), 0].item() - 1.0) < 1e-6                   # highest frequency = 1 rad per grid step
    x = torch.randn(2, cfg.heads, cfg.tokens, hd)
    y = V.rotate_pairs(x, cos, sin)
    assert torch.allclose(y.norm(dim=-1), x.norm(dim=-1), rtol=1e-5, atol=1e-5)
    # relative property along x: <R(x1) q, R(x2) k> depends on x1 - x2 only (same row)
    q, k = torch.randn(hd), torch.randn(hd)
    def dot(xa, xb, row):
        return (V.rotate_pairs(q, cos[tok(row, xa)], sin[tok(row, xa)]) *
                V.rotate_pairs(k, cos[tok(row, xb)], sin[tok(row, xb)])).sum()
    assert abs(dot(2, 6, 4) - dot(10, 14, 4)) < 1e-4


def test_full_tower_runs_and_normalises():
    cfg = PE_VISION_CONFIGS["pe-tiny"]
    sd = init_vision_state_dict(cfg, seed=5)
    x = torch.randn(3, 3, cfg.image_size, cfg.image_size)
    f = V.encode_image(sd, cfg, x, normalize=True)
    assert f.shape == (3, cfg.output_dim)
    assert torch.allclose(f.norm(dim=-1), torch.ones(3), atol=1e-5)
    # rope on/off and pooling type both change the result (every branch is live on seeded weights)
    f2 = V.encode_image(sd, dataclasses.replace(cfg, use_rope2d=False), x, normalize=True)
    f3 = V.encode_image(sd, dataclasses.replace(cfg, pool_type="tok"), x, normalize=True)
    assert (f - f2).abs().max() > 1e-3 and (f - f3).abs().max() > 1e-3
