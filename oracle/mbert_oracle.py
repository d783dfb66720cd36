"""CPU oracle for the ModernBERT text tower of the Judge reranker / PE-A-Frame span predictor (SURVEY.md section 8 rows
a17 / a18)  --  TEST INFRASTRUCTURE ONLY (same rules as oracle/samaudio_oracle.py).

What is restated here
---------------------
Reference sam_audio/model/judge.py:48 builds `AutoModel.from_config(ModernBertConfig(**cfg.text_model))` and :74-88 takes
`self.text_model(input_ids=..., attention_mask=..., output_hidden_states=True).hidden_states[nth_text_layer]`.  The
algorithm lives in the third-party package `transformers` (this image: 5.15.0, models/modernbert/modeling_modernbert.py).
Its published forward, restated functionally on a `ModernBertModel.state_dict()`:

* `h = LayerNorm(tok_embeddings[input_ids])` (LayerNorm without bias, eps = norm_eps);
* per layer i: `n = LayerNorm_i(h)` (layer 0: identity); `q, k, v = Wqkv n` split as [3, heads, head_dim];
  rotary embedding in the rotate-half convention (`x cos + (-x2, x1) sin`, `cos / sin = f(cat(freqs, freqs))`,
  `freqs = pos / theta^(2i / head_dim)`) with theta = global_rope_theta on the layers with `i % global_attn_every_n_layers
  == 0` and local_rope_theta on the others; `softmax(q k^T / sqrt(head_dim) + mask)` in fp32 where the mask adds
  finfo.min on padding keys and - on the local layers - on keys further than local_attention / 2 tokens away;
  `h = h + Wo (softmax v)`; `h = h + Wo_mlp(gelu(input) * gate)` with `input, gate = Wi LayerNorm(h)` (erf GELU);
* `last_hidden_state = LayerNorm_final(h)`; `hidden_states[n]` = h after n layers for n < num_hidden_layers (0 = the
  embeddings); `hidden_states[num_hidden_layers]` IS `last_hidden_state` (transformers 5.x records the normalised tensor).

Pinning
-------
PINNED: tests/test_mbert_oracle_cpu.py checks every hidden state and the last_hidden_state of this restatement against
`transformers.ModernBertModel` itself (the reference's own dependency) on seeded weights - ModernBERT-base dims and small
ones, ragged masks, sequences longer than the local window.  The GPU tests additionally compare the HIP path with the
module directly.
"""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _rope(cfg: Any, theta: float, L: int) -> Tuple[Tensor, Tensor]:
    hd = cfg.hidden_size // cfg.num_attention_heads
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    freqs = torch.arange(L, dtype=torch.float)[:, None] * inv[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def _rot(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    half = x.shape[-1] // 2
    return x * cos + torch.cat((-x[..., half:], x[..., :half]), dim=-1) * sin


def mbert_hidden_states(sd: Dict[str, Tensor], cfg: Any, input_ids: Tensor, attention_mask: Tensor,
                        last_prenorm: bool = False) -> Tuple[List[Tensor], Tensor]:
    """-> (hidden_states[0 .. num_hidden_layers], last_hidden_state).  `last_prenorm`: what hidden_states[num_hidden_layers]
    is - False = transformers 5.x (the normalised tensor, what the installed module returns and the default pin of
    tests/test_mbert_oracle_cpu.py), True = transformers 4.48 - 4.5x (the last layer's output before final_norm: the
    reference's pinned generation, pinned here against the INPUT of the installed module's final_norm).
    cfg: any object with the fields hidden_size,
    num_attention_heads, num_hidden_layers, global_attn_every_n_layers, local_attention, norm_eps, global_rope_theta,
    local_rope_theta (transformers 5 keeps the last four inside `layer_types` / `rope_parameters`)."""
    global_theta, local_theta = cfg.global_rope_theta, cfg.local_rope_theta
    sd = {k: v.float() for k, v in sd.items()}
    D, H = cfg.hidden_size, cfg.num_attention_heads
    hd, eps = D // H, cfg.norm_eps
    B, L = input_ids.shape
    ln = lambda x, w: F.layer_norm(x, (D,), w, None, eps)   # noqa: E731
    h = ln(sd["embeddings.tok_embeddings.weight"][input_ids], sd["embeddings.norm.weight"])
    neg = torch.finfo(torch.float32).min
    pos = torch.arange(L)
    pad = torch.zeros(B, 1, 1, L).masked_fill(attention_mask[:, None, None, :] == 0, neg)
    far = (pos[None, :] - pos[:, None]).abs() > cfg.local_attention // 2
    masks = {True: pad.expand(B, 1, L, L), False: pad.expand(B, 1, L, L).masked_fill(far[None, None], neg)}
    ropes = {True: _rope(cfg, global_theta, L), False: _rope(cfg, local_theta, L)}
    states = [h]
    for i in range(cfg.num_hidden_layers):
        p, glob = f"layers.{i}.", i % cfg.global_attn_every_n_layers == 0
        n = h if i == 0 else ln(h, sd[p + "attn_norm.weight"])
        qkv = (n @ sd[p + "attn.Wqkv.weight"].t()).view(B, L, 3, H, hd)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
        cos, sin = ropes[glob]
        q, k = _rot(q, cos, sin), _rot(k, cos, sin)
        w = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + masks[glob], dim=-1)
        h = h + (w @ v).transpose(1, 2).reshape(B, L, D) @ sd[p + "attn.Wo.weight"].t()
        a, g = (ln(h, sd[p + "mlp_norm.weight"]) @ sd[p + "mlp.Wi.weight"].t()).chunk(2, dim=-1)
        h = h + (F.gelu(a) * g) @ sd[p + "mlp.Wo.weight"].t()
        states.append(h)
    last = ln(h, sd["final_norm.weight"])
    if not last_prenorm:   # transformers 5.x: hidden_states[layers] is the normalised tensor (= last_hidden_state);
        states[-1] = last  # 4.48 - 4.5x append the last layer's output before final_norm (states[-1] as it stands)
    return states, last
