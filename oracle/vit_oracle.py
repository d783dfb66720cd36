"""CPU oracle for the visual-prompt tower of SURVEY.md section 8 (row a4 / "next" row f3)  --  TEST INFRASTRUCTURE
ONLY (same rules as oracle/samaudio_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may import it).

What is restated here
---------------------
The reference builds `pe.CLIP.from_config("PE-Core-L14-336")` and calls `encode_image(frames, normalize=True)`
(reference sam_audio/model/vision_encoder.py:80-89).  `core.vision_encoder.pe` lives in the un-vendored
`perception_models` package (pyproject.toml:24 "@unpin-deps") and `timm` - the only other carrier of the PE-Core
ViT (`vit_pe_core_large_patch14_336`, named by the in-container HF port
transformers/models/pe_video/configuration_pe_video.py:48-54 with `global_pool = "map"`, i.e. attention pooling) -
is absent too.  The network is therefore restated from its PUBLISHED architecture:

* patch embedding `conv1` (k = stride = patch, no bias) -> [class token ;] patches -> + absolute position table
  -> `ln_pre` -> `layers` x pre-LN residual blocks { LayerNorm -> fused in_proj (q|k|v, bias) -> 2-D RoPE on q, k
  (adjacent pairs; x-frequencies on the first half of the head, y-frequencies on the second, grid coordinates
  1..G so that the class token sits at (0, 0) = identity) -> SDPA (scale head_dim^-0.5) -> out_proj (bias) ;
  LayerNorm -> c_fc (bias) -> GELU(erf) -> c_proj (bias) } -> `ln_post` on every token -> pooling
  ("attn": one learned probe attends to all tokens through nn.MultiheadAttention (8 heads), then
  x + mlp(layernorm(x)); "tok": the class token) -> `@ proj` -> optional L2 normalisation (`encode_image`).
* PE-Core-L14-336: image 336, patch 14 (24 x 24 grid + class token = 577 tokens), width 1024, 24 layers, 16 heads of
  64, MLP 4096, output 1024, abs. positions + RoPE-2D, ln_pre + ln_post, attention pooling with 8 heads.

Pinning
-------
* The blocks this network shares with CLIP's ViT (patch conv, class token, position table, pre/post LayerNorm,
  pre-LN MHA + MLP residual blocks, class-token pooling, projection) are PINNED against Hugging Face
  `CLIPVisionModelWithProjection` on seeded weights (tests/test_vit_oracle_cpu.py: `use_rope2d=False`,
  `pool_type="tok"`, strict key-for-key weight mapping).
* The attention pooling head is PINNED against torch's own `nn.MultiheadAttention` + `nn.LayerNorm` modules.
* RoPE-2D (frequency layout, the +1 grid offset, adjacent-pair rotation) follows the public description only:
  PARITY UNPINNED against perception_models.

State-dict keys are the ones `pe.CLIP`'s vision tower uses under `visual.` (`conv1.weight`, `class_embedding`,
`positional_embedding`, `ln_pre.*`, `transformer.resblocks.{i}.{ln_1,ln_2,attn.in_proj_weight,attn.in_proj_bias,
attn.out_proj,mlp.c_fc,mlp.c_proj}`, `ln_post.*`, `attn_pool.{probe,attn.*,layernorm.*,mlp.*}`, `proj`).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def rope2d_tables(cfg, grid_h: int, grid_w: int):
    """cos / sin [S, head_dim] of the 2-D rotary embedding (class token first, angle 0)."""
    hd = cfg.width // cfg.heads
    half = hd // 2                                  # channels per axis
    nfreq = half // 2                               # distinct frequencies per axis (pairs share one)
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, 2, dtype=torch.float32)[:nfreq] / half))
    off = 1 if cfg.use_cls_token else 0
    ys = torch.arange(grid_h, dtype=torch.float32) + off
    xs = torch.arange(grid_w, dtype=torch.float32) + off
    fy = (ys[:, None] * inv[None, :]).repeat_interleave(2, dim=-1)   # [gh, half], pairs (2i, 2i+1) share a frequency
    fx = (xs[:, None] * inv[None, :]).repeat_interleave(2, dim=-1)
    fy = fy[:, None, :].expand(grid_h, grid_w, half)
    fx = fx[None, :, :].expand(grid_h, grid_w, half)
    ang = torch.cat([fx, fy], dim=-1).reshape(grid_h * grid_w, hd)
    if cfg.use_cls_token:
        ang = torch.cat([torch.zeros(1, hd), ang], dim=0)
    return ang.cos(), ang.sin()


def rotate_pairs(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """out = x * cos + rotate_half(x) * sin with rotate_half over adjacent pairs: (x0, x1) -> (-x1, x0)."""
    x0, x1 = x[..., 0::2], x[..., 1::2]
    rot = torch.stack((-x1, x0), dim=-1).flatten(-2)
    return x * cos + rot * sin


def act_fn(cfg, x: Tensor) -> Tensor:
    if cfg.act == "gelu":
        return F.gelu(x)
    if cfg.act == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    raise ValueError(cfg.act)


def mha(x_q: Tensor, x_kv: Tensor, w_in: Tensor, b_in: Tensor, w_out: Tensor, b_out: Tensor, heads: int,
        rope=None) -> Tensor:
    """fused-in_proj multi-head attention: q from x_q, k / v from x_kv; optional (cos, sin) applied to q and k."""
    D = x_q.shape[-1]
    hd = D // heads
    q = F.linear(x_q, w_in[:D], b_in[:D])
    k = F.linear(x_kv, w_in[D:2 * D], b_in[D:2 * D])
    v = F.linear(x_kv, w_in[2 * D:], b_in[2 * D:])

    def split(t):
        return t.reshape(t.shape[0], t.shape[1], heads, hd).transpose(1, 2)
    q, k, v = split(q), split(k), split(v)
    if rope is not None:
        q, k = rotate_pairs(q, *rope), rotate_pairs(k, *rope)
    att = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1) @ v
    att = att.transpose(1, 2).reshape(x_q.shape[0], x_q.shape[1], D)
    return F.linear(att, w_out, b_out)


def attn_pool(sd: SD, cfg, x: Tensor) -> Tensor:
    """AttentionPooling: probe -> MHA(q = probe, k = v = x) -> x + mlp(layernorm(x)); [N, S, D] -> [N, D]."""
    p = "attn_pool."
    q = sd[p + "probe"].reshape(1, 1, -1).expand(x.shape[0], -1, -1)
    y = mha(q, x, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.weight"],
            sd[p + "attn.out_proj.bias"], cfg.attn_pooler_heads)
    z = F.layer_norm(y, (y.shape[-1],), sd[p + "layernorm.weight"], sd[p + "layernorm.bias"], cfg.ln_eps)
    z = F.linear(act_fn(cfg, F.linear(z, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])),
                 sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    return (y + z)[:, 0]


def vision_tower(sd: SD, cfg, frames: Tensor, taps: dict | None = None) -> Tensor:
    """frames [N, 3, H, W] float (already resized / normalised) -> pooled, projected features [N, output_dim]."""
    N, _, Himg, Wimg = frames.shape
    gh, gw = Himg // cfg.patch_size, Wimg // cfg.patch_size
    W = cfg.width
    x = F.conv2d(frames, sd["conv1.weight"], None, stride=cfg.patch_size)            # [N, W, gh, gw]
    x = x.permute(0, 2, 3, 1).reshape(N, gh * gw, W)
    if cfg.use_cls_token:
        x = torch.cat([sd["class_embedding"].reshape(1, 1, W).expand(N, -1, -1), x], dim=1)
    if cfg.use_abs_posemb:
        assert gh * gw + int(cfg.use_cls_token) == sd["positional_embedding"].shape[0], \
            "oracle covers the native grid only (no position-table interpolation)"
        x = x + sd["positional_embedding"][None]
    if cfg.use_ln_pre:
        x = F.layer_norm(x, (W,), sd["ln_pre.weight"], sd["ln_pre.bias"], cfg.ln_eps)
    rope = rope2d_tables(cfg, gh, gw) if cfg.use_rope2d else None
    if taps is not None:
        taps["embed"] = x
    for i in range(cfg.layers):
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (W,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], cfg.ln_eps)
        x = x + mha(h, h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], sd[p + "attn.out_proj.weight"],
                    sd[p + "attn.out_proj.bias"], cfg.heads, rope)
        h = F.layer_norm(x, (W,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], cfg.ln_eps)
        x = x + F.linear(act_fn(cfg, F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])),
                         sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        if taps is not None:
            taps[f"layer{i}"] = x
    if cfg.use_ln_post:
        x = F.layer_norm(x, (W,), sd["ln_post.weight"], sd["ln_post.bias"], cfg.ln_eps)
    if cfg.pool_type == "attn":
        x = attn_pool(sd, cfg, x)
    elif cfg.pool_type == "tok":
        x = x[:, 0]
    elif cfg.pool_type == "avg":
        x = x.mean(dim=1)
    else:
        raise ValueError(cfg.pool_type)
    if taps is not None:
        taps["pooled"] = x
    if "proj" in sd:
        x = x @ sd["proj"]
    return x


def encode_image(sd: SD, cfg, frames: Tensor, normalize: bool = False) -> Tensor:
    """pe.CLIP.encode_image as the reference calls it (vision_encoder.py:87-89)."""
    x = vision_tower(sd, cfg, frames)
    return F.normalize(x, dim=-1) if normalize else x


def tower_flops(cfg, n_frames: int) -> float:
    """algorithmic FLOPs (2 x MACs) of `n_frames` tower evaluations at the native resolution."""
    g = cfg.image_size // cfg.patch_size
    S, W = g * g + int(cfg.use_cls_token), cfg.width
    F_ = int(W * cfg.mlp_ratio)
    per = 2.0 * g * g * 3 * cfg.patch_size ** 2 * W
    per += cfg.layers * (2.0 * S * W * 3 * W + 4.0 * S * S * W + 2.0 * S * W * W + 4.0 * S * W * F_)
    if cfg.pool_type == "attn":
        per += 2.0 * S * W * 2 * W + 4.0 * S * W + 2.0 * W * W + 4.0 * W * F_
    per += 2.0 * W * cfg.output_dim
    return per * n_frames
