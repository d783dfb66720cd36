"""CPU oracle for the text-prompt encoder of SURVEY.md section 8 (row a3 / "next" row f4)  --  TEST INFRASTRUCTURE ONLY
(same rules as oracle/samaudio_oracle.py: only tests/, smoke() and bench.py's cpu_baseline leg may import it).

What is restated here
---------------------
The reference's `T5TextEncoder` (sam_audio/model/text_encoder.py:11-37) tokenises the prompts and returns
`transformers.T5EncoderModel("t5-base")(input_ids, attention_mask)["last_hidden_state"]` with the boolean mask.  The
algorithm lives in the third-party package `transformers` (pyproject dependency of the reference; this image carries
transformers 5.15.0: models/t5/modeling_t5.py `T5Stack` / `T5Block` / `T5Attention` / `T5LayerNorm` /
`T5DenseActDense`).  Its published forward, restated functionally on a `T5EncoderModel.state_dict()`:

* `h = shared[input_ids]` (no scaling, dropout is the identity in eval mode);
* per block: `n = h * rsqrt(mean(h^2) + eps) * w` (T5LayerNorm: no mean subtraction, no bias, fp32);
  `q, k, v = n Wq^T, n Wk^T, n Wv^T` split into heads of d_kv; `scores = q k^T` (NO 1/sqrt(d) factor)
  `+ position_bias + (1 - mask) * finfo.min`; `position_bias[h, i, j] = relative_attention_bias[bucket(j - i)][h]`
  with the bidirectional log-bucket rule, computed by block 0 and shared by every block; softmax in fp32;
  `h = h + (softmax v) Wo^T`;  `n = T5LayerNorm(h)`; `h = h + act(n Wi^T) Wo^T` (act = ReLU for t5-base);
* `last_hidden_state = T5LayerNorm_final(h)` at EVERY position, padding rows included.

Pinning
-------
PINNED: tests/test_t5_oracle_cpu.py checks this restatement against `transformers.T5EncoderModel` itself (the
reference's own dependency, present in this image) on seeded weights - t5-base dims and small ones, ragged masks,
ReLU and gelu_new - to fp32 summation-order noise.  The GPU tests additionally compare the HIP path with
`T5EncoderModel` directly.
"""
from __future__ import annotations

import math
from typing import Any, Dict

import torch

Tensor = torch.Tensor


def _bucket(rel: Tensor, num_buckets: int, max_distance: int) -> Tensor:
    """modeling_t5.py T5Attention._relative_position_bucket, bidirectional=True."""
    nb = num_buckets // 2
    out = (rel > 0).long() * nb
    dist = rel.abs()
    exact = nb // 2
    far = exact + (torch.log(dist.float() / exact) / math.log(max_distance / exact) * (nb - exact)).long()
    far = far.clamp(max=nb - 1)
    return out + torch.where(dist < exact, dist, far)


def t5_layer_norm(x: Tensor, w: Tensor, eps: float) -> Tensor:
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def _act(name: str, x: Tensor) -> Tensor:
    if name == "relu":
        return torch.relu(x)
    if name == "gelu_new":
        return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))
    raise NotImplementedError(name)


def t5_encoder(sd: Dict[str, Tensor], cfg: Any, input_ids: Tensor, attention_mask: Tensor, taps=None) -> Tensor:
    """cfg: any object with the T5Config fields d_kv, num_heads, num_layers, relative_attention_num_buckets,
    relative_attention_max_distance, layer_norm_epsilon, dense_act_fn.  Returns last_hidden_state [B, L, d_model] f32."""
    H, dkv, eps = cfg.num_heads, cfg.d_kv, cfg.layer_norm_epsilon
    B, L = input_ids.shape
    sd = {k: v.float() for k, v in sd.items()}
    h = sd["shared.weight"][input_ids]
    pos = torch.arange(L)
    rel = pos[None, :] - pos[:, None]                                        # key - query
    table = sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    bias = table[_bucket(rel, cfg.relative_attention_num_buckets, cfg.relative_attention_max_distance)]  # [L, L, H]
    bias = bias.permute(2, 0, 1)[None]                                       # [1, H, L, L]
    neg = (1.0 - attention_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    bias = bias + neg
    for i in range(cfg.num_layers):
        p = f"encoder.block.{i}.layer."
        n = t5_layer_norm(h, sd[p + "0.layer_norm.weight"], eps)
        split = lambda t: t.view(B, L, H, dkv).transpose(1, 2)               # noqa: E731
        q = split(n @ sd[p + "0.SelfAttention.q.weight"].t())
        k = split(n @ sd[p + "0.SelfAttention.k.weight"].t())
        v = split(n @ sd[p + "0.SelfAttention.v.weight"].t())
        w = torch.softmax(q @ k.transpose(-1, -2) + bias, dim=-1)
        a = (w @ v).transpose(1, 2).reshape(B, L, H * dkv)
        h = h + a @ sd[p + "0.SelfAttention.o.weight"].t()
        n = t5_layer_norm(h, sd[p + "1.layer_norm.weight"], eps)
        u = _act(cfg.dense_act_fn, n @ sd[p + "1.DenseReluDense.wi.weight"].t())
        h = h + u @ sd[p + "1.DenseReluDense.wo.weight"].t()
        if taps is not None:
            taps[f"layer{i}"] = h.clone()
    return t5_layer_norm(h, sd["encoder.final_layer_norm.weight"], eps)
