"""CPU oracle for the SAMAudio.separate() hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch fp32 *restatement* of the reference algorithm.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
(sam_audio_amd/) never does and fails loudly when the HIP library is missing.

Pinning status
--------------
* DiT / patcher / RoPE / align / model glue: PINNED.  oracle/gen_golden.py runs the reference's
  own classes (imported read-only from /root/reference under a synthetic parent package) on
  seeded weights and checks this restatement against them (tests/golden/*.npz are the committed
  outputs; tests/test_oracle_golden.py re-checks the restatement against them on every run).
* Fixed-grid ODE (torchdiffeq, un-vendored, unpinned in pyproject.toml:29): restated from its
  published algorithm (fixed grid, midpoint / euler); anchored on the reference call site
  model.py:22,285-290.  PARITY UNPINNED (no reference-side fixture exists).
* DAC-VAE (dacvae, un-vendored, unpinned in pyproject.toml:18): restated from the HF `dac`
  topology; anchored on the reference call sites codec.py:45-89.  PINNED numerically against the
  two in-container implementations of the same networks (tests/test_dac_pin_cpu.py, strict
  weight-for-weight key mapping): the encoder against `PeAudioDacEncoder` (the DAC-VAE encoder copy
  inside the HF port of PE-AV, whose defaults equal the reference's DACVAEConfig), the decoder
  against `DacDecoder`.  PARITY UNPINNED against the `dacvae` package itself (not reachable
  offline: weight-norm spelling, any extra head).

Every function cites the reference lines it follows.  State-dict keys are the reference's.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# small pieces
# ----------------------------------------------------------------------------------------------
def rms_norm(x: Tensor, weight: Tensor, eps: float) -> Tensor:
    """transformer.py:36-47 - fp32 inside, eps inside the rsqrt, weight applied after."""
    xf = x.float()
    inv = torch.rsqrt((xf * xf).mean(dim=-1, keepdim=True) + eps)
    return (xf * inv * weight).to(x.dtype)


def swiglu_mlp(x: Tensor, w1: Tensor, w3: Tensor, w2: Tensor) -> Tensor:
    """transformer.py:69-80 (ProjectionLayer) and :195-206 (FeedForward); dropout is inactive
    under eval()."""
    return F.linear(F.silu(F.linear(x, w1)) * F.linear(x, w3), w2)


def sincos_embedding(pos: Tensor, dim: int, theta: float = 10000.0) -> Tensor:
    """model.py:25-42 (SinusoidalEmbedding, dim=D) and transformer.py:228-248
    (TimestepEmbedder.timestep_embedding, dim=256): cat(cos, sin) of pos * theta^(-i/half).
    Both call sites feed the raw t in [0,1] (quirk Q7)."""
    half = dim // 2
    inv_freq = torch.exp(-math.log(theta) * torch.arange(half, dtype=torch.float32) / half)
    ang = pos.float()[:, None] * inv_freq[None, :]
    return torch.cat([ang.cos(), ang.sin()], dim=-1)


def rope_tables(head_dim: int, n_pos: int, theta: float) -> Tuple[Tensor, Tensor]:
    """rope.py:116-145: freq_i = theta^(-2i/head_dim), angle = pos*freq, fp32."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(n_pos), freqs).float()
    return ang.cos(), ang.sin()


def apply_rope(x: Tensor, cos: Tensor, sin: Tensor) -> Tensor:
    """rope.py:147-155 with the [[cos,-sin],[sin,cos]] matrix of :145 - adjacent pairs (2i,2i+1);
    x is [B, H, L, E]."""
    L = x.shape[2]
    x0, x1 = x[..., 0::2], x[..., 1::2]
    c, s = cos[:L][None, None], sin[:L][None, None]
    out = torch.stack([x0 * c - x1 * s, x0 * s + x1 * c], dim=-1)
    return out.flatten(-2).to(x.dtype)


def split_heads_interleaved(x: Tensor, n_heads: int) -> Tensor:
    """transformer.py:121-126: channel c = d*H + h (head index is the fast axis)."""
    B, T, C = x.shape
    return x.reshape(B, T, C // n_heads, n_heads).permute(0, 3, 1, 2)


def attention(sd: SD, prefix: str, x: Tensor, n_heads: int, eps: float, *,
              cross: Optional[Tensor] = None, key_mask: Optional[Tensor] = None,
              rope: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """transformer.py:128-161.  qk-norm weight is shared by all heads (:117-119); RoPE after the
    norm and for self-attention only; SDPA default scale 1/sqrt(head_dim); bool mask True=attend."""
    src = x if cross is None else cross
    q = split_heads_interleaved(F.linear(x, sd[prefix + "wq.weight"]), n_heads)
    k = split_heads_interleaved(F.linear(src, sd[prefix + "wk.weight"]), n_heads)
    v = split_heads_interleaved(F.linear(src, sd[prefix + "wv.weight"]), n_heads)
    q = rms_norm(q, sd[prefix + "q_norm.weight"], eps)
    k = rms_norm(k, sd[prefix + "k_norm.weight"], eps)
    if rope is not None:
        q, k = apply_rope(q, *rope), apply_rope(k, *rope)
    scores = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(q.shape[-1])
    if key_mask is not None:
        scores = scores.masked_fill(~key_mask[:, None, None, :], float("-inf"))
    out = torch.matmul(torch.softmax(scores, dim=-1), v)  # [B,H,T,hd]
    B, H, T, hd = out.shape
    out = out.permute(0, 2, 1, 3).reshape(B, T, H * hd)  # head-major merge, transformer.py:160
    return F.linear(out, sd[prefix + "wo.weight"])


def group_norm_1(x_btc: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """patcher.py:83-85 with num_groups=1 (:155-159): statistics over (C x T) per sample,
    padded frames included (quirk Q5).  Channels-last input here."""
    mean = x_btc.mean(dim=(1, 2), keepdim=True)
    var = x_btc.var(dim=(1, 2), unbiased=False, keepdim=True)
    return (x_btc - mean) * torch.rsqrt(var + eps) * weight + bias


def conv3_same(x_btc: Tensor, weight: Tensor, bias: Tensor) -> Tensor:
    """patcher.py:53-67 for k=3, stride 1, dilation 1: padding_total=2 -> (1,1), zero pad."""
    return F.conv1d(x_btc.transpose(1, 2), weight, bias, padding=1).transpose(1, 2)


def patcher(sd: SD, prefix: str, x: Tensor) -> Tensor:
    """patcher.py:138-141,161-164 with patch_size=1 and Identity skip (quirk Q6)."""
    h = x
    for blk in ("block1", "block2"):
        p = f"{prefix}block.{blk}."
        h = group_norm_1(h, sd[p + "groupnorm.weight"], sd[p + "groupnorm.bias"])
        h = conv3_same(F.silu(h), sd[p + "project.weight"], sd[p + "project.bias"])
    return h + x


# ----------------------------------------------------------------------------------------------
# DiT
# ----------------------------------------------------------------------------------------------
def dit_block(sd: SD, p: str, x: Tensor, y: Tensor, t0: Tensor, n_heads: int, eps: float,
              pad_mask, mem_mask, rope) -> Tensor:
    """transformer.py:354-391."""
    B, _, D = x.shape
    mod = sd[p + "scale_shift_table"][None] + t0.reshape(B, 6, D)
    sh1, sc1, g1, sh2, sc2, g2 = [mod[:, i : i + 1] for i in range(6)]
    a = rms_norm(x, sd[p + "attention_norm.weight"], eps) * (1 + sc1) + sh1
    h = x + g1 * attention(sd, p + "attention.", a, n_heads, eps, key_mask=pad_mask, rope=rope)
    h = h + attention(sd, p + "cross_attention.", h, n_heads, eps, cross=y, key_mask=mem_mask)
    f = rms_norm(h, sd[p + "ffn_norm.weight"], eps) * (1 + sc2) + sh2
    ff = swiglu_mlp(f, sd[p + "feed_forward.w1.weight"], sd[p + "feed_forward.w3.weight"],
                    sd[p + "feed_forward.w2.weight"])
    return h + g2 * ff


def dit_forward(sd: SD, cfg, x: Tensor, time: Tensor, *, pad_mask=None, memory=None,
                mem_mask=None, prefix: str = "transformer.", taps: Optional[dict] = None) -> Tensor:
    """transformer.py:473-524.  `cfg` is a TransformerConfig; `taps` collects per-stage tensors."""
    P = prefix
    eps = cfg.norm_eps
    h = patcher(sd, P + "x_embedder.", x)
    if taps is not None:
        taps["patcher"] = h
    temb = sincos_embedding(time, cfg.frequency_embedding_dim)
    t = swiglu_mlp(temb, sd[P + "t_embedder.projection.w1.weight"],
                   sd[P + "t_embedder.projection.w3.weight"], sd[P + "t_embedder.projection.w2.weight"])
    t0 = F.linear(F.silu(t), sd[P + "t_block.weight"], sd[P + "t_block.bias"])
    y = swiglu_mlp(memory, sd[P + "y_embedder.projection.w1.weight"],
                   sd[P + "y_embedder.projection.w3.weight"], sd[P + "y_embedder.projection.w2.weight"])
    rope = rope_tables(cfg.dim // cfg.n_heads, x.shape[1], float(max(10000, 2 * cfg.max_positions)))
    for i in range(cfg.n_layers):
        h = dit_block(sd, f"{P}layers.{i}.", h, y, t0, cfg.n_heads, eps, pad_mask, mem_mask, rope)
        if taps is not None:
            taps[f"layer{i}"] = h
    mod = sd[P + "final_layer_scale_shift_table"][None] + t[:, None]
    shift, scale = mod[:, 0:1], mod[:, 1:2]
    h = rms_norm(h, sd[P + "norm.weight"], eps) * (1 + scale) + shift
    return F.linear(h, sd[P + "output.weight"])


# ----------------------------------------------------------------------------------------------
# SAMAudio glue
# ----------------------------------------------------------------------------------------------
def align_inputs(sd: SD, noisy: Tensor, feats: Tensor, video: Optional[Tensor],
                 anchor_ids: Optional[Tensor], anchor_alignment: Optional[Tensor]) -> Tensor:
    """model.py:108-128, align.py:30-50, model.py:54-65.  `video` is [B, Cv, T]."""
    x = torch.cat([noisy, torch.zeros_like(feats), feats], dim=2)
    out = F.linear(x, sd["proj.weight"], sd["proj.bias"])
    if video is not None:
        pc = F.conv1d(video, sd["align_masked_video.conv.weight"], sd["align_masked_video.conv.bias"])
        pc = F.layer_norm(pc.transpose(1, 2), (pc.shape[1],), sd["align_masked_video.layer_norm.weight"],
                          sd["align_masked_video.layer_norm.bias"], 1e-5)
        out = out + torch.tanh(sd["align_masked_video.gate"]) * pc
    if anchor_ids is not None:
        tok = anchor_ids.gather(1, anchor_alignment)
        emb = F.embedding(tok, sd["embed_anchors.embed.weight"])
        out = out + torch.tanh(sd["embed_anchors.gate"]) * F.linear(emb, sd["embed_anchors.proj.weight"])
    return out


def samaudio_forward(sd: SD, cfg, noisy: Tensor, feats: Tensor, text: Optional[Tensor], time: Tensor,
                     video=None, text_mask=None, anchor_ids=None, anchor_alignment=None,
                     pad_mask=None, taps: Optional[dict] = None) -> Tensor:
    """model.py:130-180 - one ODE function evaluation.  `cfg` is a SAMAudioConfig."""
    x = align_inputs(sd, noisy, feats, video, anchor_ids, anchor_alignment)
    if taps is not None:
        taps["aligned"] = x
    tmem = sincos_embedding(time, cfg.transformer.dim)[:, None]
    memory = tmem
    if text is not None:
        memory = F.linear(text, sd["memory_proj.weight"], sd["memory_proj.bias"]) + tmem
    return dit_forward(sd, cfg.transformer, x, time, pad_mask=pad_mask, memory=memory,
                       mem_mask=text_mask, taps=taps)


def ode_fixed_grid(fn, y0: Tensor, method: str = "midpoint", step_size: float = 2 / 32,
                   t0: float = 0.0, t1: float = 1.0, record: Optional[list] = None) -> Tensor:
    """Restatement of torchdiffeq's fixed-grid solvers as used at model.py:285-290 with
    DFLT_ODE_OPT (model.py:22): grid = t0 + k*step (last point clamped to t1);
    midpoint: y += dt * f(t + dt/2, y + dt/2 * f(t, y));  euler: y += dt * f(t, y)."""
    if method not in ("midpoint", "euler"):
        raise ValueError(f"unsupported ODE method {method!r}")
    n = int(math.ceil((t1 - t0) / step_size + 1))
    grid = [min(t0 + k * step_size, t1) for k in range(n)]
    grid[-1] = t1
    y = y0
    for ta, tb in zip(grid[:-1], grid[1:]):
        dt = tb - ta
        k1 = fn(torch.tensor(ta, dtype=torch.float32), y)
        if method == "euler":
            y = y + dt * k1
        else:
            y = y + dt * fn(torch.tensor(ta + 0.5 * dt, dtype=torch.float32), y + (0.5 * dt) * k1)
        if record is not None:
            record.append(y)
    return y


# ----------------------------------------------------------------------------------------------
# DAC-VAE (HF `dac` topology; channels-first like the reference)
# ----------------------------------------------------------------------------------------------
def snake(x: Tensor, alpha: Tensor) -> Tensor:
    """transformers/models/dac/modeling_dac.py:86-101."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def _res_unit(sd: SD, p: str, x: Tensor, dilation: int) -> Tensor:
    """modeling_dac.py:175-209: snake, conv k7 (dilated, 'same'), snake, conv k1, residual."""
    h = F.conv1d(snake(x, sd[p + "block.0.alpha"]), sd[p + "block.1.weight"], sd[p + "block.1.bias"],
                 dilation=dilation, padding=3 * dilation)
    h = F.conv1d(snake(h, sd[p + "block.2.alpha"]), sd[p + "block.3.weight"], sd[p + "block.3.bias"])
    return x + h


def dac_pad(wav: Tensor, hop: int) -> Tensor:
    """codec.py:72-78: right reflect-pad to a multiple of the hop length (quirk Q15)."""
    rem = wav.shape[-1] % hop
    return wav if rem == 0 else F.pad(wav, (0, hop - rem), mode="reflect")


def dac_encode(sd: SD, cfg, wav: Tensor, prefix: str = "audio_codec.") -> Tensor:
    """codec.py:65-70: encoder -> quantizer.in_proj -> chunk(2) keep the mean.  wav [B,1,Tw]
    -> [B, codebook_dim, T].  Encoder topology modeling_dac.py:212-233,444-474."""
    E = prefix + "encoder.block."
    x = F.conv1d(dac_pad(wav, cfg.hop_length), sd[E + "0.weight"], sd[E + "0.bias"], padding=3)
    for i, s in enumerate(cfg.encoder_rates):
        B = f"{E}{i + 1}.block."
        for j, d in enumerate((1, 3, 9)):
            x = _res_unit(sd, f"{B}{j}.", x, d)
        x = snake(x, sd[B + "3.alpha"])
        x = F.conv1d(x, sd[B + "4.weight"], sd[B + "4.bias"], stride=s, padding=math.ceil(s / 2))
    x = snake(x, sd[E + "5.alpha"])
    x = F.conv1d(x, sd[E + "6.weight"], sd[E + "6.bias"], padding=1)
    q = F.conv1d(x, sd[prefix + "quantizer.in_proj.weight"], sd[prefix + "quantizer.in_proj.bias"])
    return q.chunk(2, dim=1)[0]


def dac_decode(sd: SD, cfg, z: Tensor, prefix: str = "audio_codec.") -> Tensor:
    """codec.py:86-89: quantizer.out_proj -> decoder.  z [N, codebook_dim, T] -> [N,1,T*hop].
    Decoder topology modeling_dac.py:236-264,407-441."""
    x = F.conv1d(z, sd[prefix + "quantizer.out_proj.weight"], sd[prefix + "quantizer.out_proj.bias"])
    Dm = prefix + "decoder.model."
    x = F.conv1d(x, sd[Dm + "0.weight"], sd[Dm + "0.bias"], padding=3)
    for i, s in enumerate(cfg.decoder_rates):
        B = f"{Dm}{i + 1}.block."
        x = snake(x, sd[B + "0.alpha"])
        x = F.conv_transpose1d(x, sd[B + "1.weight"], sd[B + "1.bias"], stride=s, padding=math.ceil(s / 2))
        for j, d in enumerate((1, 3, 9)):
            x = _res_unit(sd, f"{B}{j + 2}.", x, d)
    x = snake(x, sd[Dm + "5.alpha"])
    x = F.conv1d(x, sd[Dm + "6.weight"], sd[Dm + "6.bias"], padding=3)
    return torch.tanh(x)


# ----------------------------------------------------------------------------------------------
# host-side integer work (processor.py) and the separate() assembly
# ----------------------------------------------------------------------------------------------
def anchors_to_ids(anchors, pad_mask: Tensor, hop: int, sample_rate: int):
    """processor.py:78-124 (Batch.process_anchors).  Returns (anchor_ids, anchor_alignment)."""
    vocab = {"<null>": 0, "+": 1, "-": 2, "<pad>": 3}
    B, T = pad_mask.shape
    align = torch.zeros(B, T, dtype=torch.long)
    align[~pad_mask] = 1
    if anchors is None:
        ids = torch.zeros(B, 2, dtype=torch.long)
        ids[:, 1] = vocab["<pad>"]
        return ids, align
    rows: List[Tensor] = []
    for b, lst in enumerate(anchors):
        cur = [vocab["<null>"], vocab["<pad>"]]
        for tok, t_start, t_end in lst:
            s = math.ceil(t_start * sample_rate / hop)
            e = math.ceil(t_end * sample_rate / hop)
            align[b, s:e] = len(cur)
            cur.append(vocab[tok])
        rows.append(torch.tensor(cur))
    ids = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=vocab["<pad>"])
    return ids, align


def separate(sd: SD, cfg, audios: Tensor, sizes: Tensor, text: Tensor, text_mask: Tensor,
             noise: Tensor, *, anchors=None, video: Optional[Tensor] = None, candidates: int = 1,
             method: str = "midpoint", step_size: float = 2 / 32, decode: bool = True,
             record: Optional[list] = None):
    """model.py:247-338 without rerankers (candidate 0 is returned, the `else` arm of :329-330).
    audios [B,1,Tw] fp32, sizes [B] latent frames, text [B,Lt,768], noise [B*cand,T,256].
    Returns (target list, residual list, final latent [B*cand,T,256])."""
    codec = cfg.audio_codec
    z = dac_encode(sd, codec, audios).transpose(1, 2)          # model.py:182-184
    feats = torch.cat([z, z], dim=2)
    B, T, _ = feats.shape
    pad_mask = torch.arange(T)[None, :] < sizes[:, None]       # processor.py:127-128
    ids, align = anchors_to_ids(anchors, pad_mask, codec.hop_length, codec.sample_rate)
    if video is None:
        video = feats.new_zeros(B, cfg.vision_encoder.dim, T)  # model.py:186-189 (quirk Q8)

    def rep(x):                                                 # model.py:193-203
        return x if candidates == 1 else x.repeat_interleave(candidates, dim=0)

    feats_r, text_r, tmask_r, video_r = rep(feats), rep(text), rep(text_mask), rep(video)
    ids_r, align_r, pad_r = rep(ids), rep(align), rep(pad_mask)

    def field_fn(t, y):
        return samaudio_forward(sd, cfg, y, feats_r, text_r, t.expand(y.shape[0]), video=video_r,
                                text_mask=tmask_r, anchor_ids=ids_r, anchor_alignment=align_r,
                                pad_mask=pad_r)

    latent = ode_fixed_grid(field_fn, noise, method=method, step_size=step_size, record=record)
    if not decode:
        return None, None, latent
    Bc = latent.shape[0]
    C = latent.shape[2] // 2
    gen = latent.transpose(1, 2).reshape(2 * Bc, C, T)          # model.py:291-295
    wavs = dac_decode(sd, codec, gen).view(Bc, 2, -1)
    wav_sizes = (sizes * codec.hop_length).tolist()             # codec.py:91-97
    target = [wavs[b * candidates, 0, : wav_sizes[b]] for b in range(B)]
    residual = [wavs[b * candidates, 1, : wav_sizes[b]] for b in range(B)]
    return target, residual, latent
