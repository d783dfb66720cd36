"""Reference state_dict -> engine tensors.

The checkpoint format is the reference's flat ``state_dict`` (``checkpoint.pt``, reference
sam_audio/model/base.py:56-61; key names enumerated in SURVEY.md §8b).  The HIP engine wants a few
one-time re-layouts, all done here on the weights' device:

* **Q1 interleaved heads** (reference transformer.py:121-126): q/k/v projection channel ``c = d*H + h``.
  Rows of wq/wk/wv are permuted to head-major (``h*128 + d``) so a 128-wide column block of the QKV
  GEMM is one head; ``wo`` already consumes head-major channels (transformer.py:160).
* wq|wk|wv fused into one [3D, D] operand; cross-attention wk|wv fused into [2D, D].
* SwiGLU ``w1``/``w3`` interleaved in 16-row blocks so the GEMM epilogue finds silu(w1 x) and w3 x of
  the same column in one lane (transformer.py:195-206, :69-80).
* **Q10**: columns 256..511 of ``proj`` multiply zeros (model.py:116-123) and are dropped; the other two
  blocks are split into ``proj_wy`` (noisy audio) and ``proj_wf`` (audio features, hoisted out of the ODE).
* Conv1d weights [Cout, Cin, k] -> [Cout, k*Cin] (tap-major K, channels-last activations);
  ConvTranspose1d [Cin, Cout, 2s] -> [s*Cout, 2*Cin] (phase-major rows);
  K zero-padded to the GEMM slab (64 bf16 / 32 f32 elements); weight-norm (g, v) pairs are folded.
* tanh(gate) of the anchor term is folded into ``anc_w`` (model.py:65).
* RoPE / sinusoid tables are computed once on the CPU with the reference's own op sequence
  (rope.py:116-145, model.py:27-33, transformer.py:228-234) and kept in fp32 (quirk Q17).
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Tuple

import torch

from .config import SAMAudioConfig

# keys the reference checkpoint does not carry (reference model.py:351-355)
OPTIONAL_KEY_RE = re.compile(r"(^text_encoder|^visual_ranker|^text_ranker|^span_predictor)")
# keys a genuine reference checkpoint DOES carry but the separate() engine does not consume: the reference SAMAudio owns
# `self.vision_encoder = PerceptionEncoder(...)` (model.py:83, a pe.CLIP under `vision_encoder.model.*`) and its strict
# load forgives only the four prefixes above, so checkpoint.pt holds those tensors.  They are routed to
# `model.vision_encoder.load_state_dict` when a tower with that method is attached and are otherwise ignored - never
# reported as unexpected.
IGNORED_KEY_RE = re.compile(r"^vision_encoder\.")


def _head_major(w: torch.Tensor, n_heads: int) -> torch.Tensor:
    out_f, in_f = w.shape
    hd = out_f // n_heads
    return w.reshape(hd, n_heads, in_f).permute(1, 0, 2).reshape(out_f, in_f)


def _interleave16(w1: torch.Tensor, w3: torch.Tensor) -> torch.Tensor:
    f, k = w1.shape
    assert f % 16 == 0
    return torch.stack([w1.reshape(f // 16, 16, k), w3.reshape(f // 16, 16, k)], dim=1).reshape(2 * f, k)


def ktm_layout(w: torch.Tensor) -> torch.Tensor:
    """[N, K] row-major -> [K/64, N, 64] K-TILE-MAJOR (include/samaudio.h): the 64-element K slab of all N rows contiguous,
    so that a GEMM launch walks the weight matrix front to back - one contiguous N*128-byte run per K-tile instead of a
    128-byte piece out of each of N rows 2K bytes apart (what a few-row launch waits for with cold weights, DESIGN.md
    section 7).  The 8-phase GEMM family reads either layout into the same LDS image: same results."""
    n, k = w.shape
    assert k % 64 == 0
    return w.reshape(n, k // 64, 64).permute(1, 0, 2).contiguous()


def ktm_to_rows(w: torch.Tensor) -> torch.Tensor:
    """inverse of ktm_layout: [K/64, N, 64] -> [N, K]"""
    kt, n, s = w.shape
    return w.permute(1, 0, 2).reshape(n, kt * s).contiguous()


def x3_weight(w: torch.Tensor, half: torch.dtype, ktm: bool = True) -> torch.Tensor:
    """fp32 [N, K] -> the split operand [W_hi | W_lo | W_hi] of a compensated 16-bit GEMM (include/samaudio.h
    SAMAUDIO_OPT_X3_CLASSES): W_hi = rn16(W), W_lo = rn16(W - W_hi), so W = W_hi + W_lo to ~2^-22 (IEEE half) and the activation
    row [x_lo | x_hi | x_hi] gives x_lo W_hi + x_hi W_lo + x_hi W_hi in one pass over K' = 3K.  [N, 3K] row-major, or K-tile-major
    [3K/64, N, 64] (`ktm`, as ktm_layout)."""
    w = w.float()
    hi = w.to(half)
    if half == torch.float16:   # a weight beyond the format's range would split into inf - inf
        hi = w.clamp(-65504.0, 65504.0).to(half)
    lo = (w - hi.float()).to(half)
    w3 = torch.cat([hi, lo, hi], dim=1).contiguous()
    return ktm_layout(w3) if ktm else w3


def _pad_k(w: torch.Tensor, slab: int) -> torch.Tensor:
    n, k = w.shape
    kp = (k + slab - 1) // slab * slab
    if kp == k:
        return w
    out = w.new_zeros(n, kp)
    out[:, :k] = w
    return out


def _conv_weight(sd: Dict[str, torch.Tensor], name: str, transposed: bool = False) -> torch.Tensor:
    """Plain `weight`, or weight-norm in either the legacy (weight_g / weight_v) or the parametrised
    (parametrizations.weight.original0/1) spelling; norm over all dims but 0 (torch weight_norm default)."""
    if name + ".weight" in sd:
        return sd[name + ".weight"].float()
    for g_key, v_key in ((".weight_g", ".weight_v"),
                         (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
        if name + g_key in sd:
            g, v = sd[name + g_key].float(), sd[name + v_key].float()
            norm = v.flatten(1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            return g * v / norm
    raise KeyError(name + ".weight")


def expected_keys(cfg: SAMAudioConfig, with_codec: bool = True) -> List[str]:
    from .synthetic import init_state_dict  # key enumeration only (meta device: no memory)
    return list(init_state_dict(cfg, device="meta", with_codec=with_codec).keys())


KTM_LEAVES = ("wqkv", "wo", "c_wq", "w13", "w2")   # per-layer weights that may be stored K-tile-major
# the GEMM classes a 16-bit engine can run on exact-fp32 operands (hip.CLS_F32_CAPABLE) and the engine weights each reads
F32_CLASS_WEIGHTS = {"out": ("w_out",), "time": ("t_w13", "t_w2", "tb_w"), "in": ("proj_wy",),
                     "prep": ("proj_wf", "mem_w", "vid_w", "anc_w"), "yemb": ("y_w13", "y_w2")}
_F32_WEIGHT_CLASS = {w: c for c, ws in F32_CLASS_WEIGHTS.items() for w in ws}
# the checkpoint entries those weights are made of (SAMAudio keeps references to them so that set_f32_classes can add a
# class's fp32 copies later without the whole checkpoint)
F32_SOURCE_KEYS = tuple(["transformer.output.weight", "transformer.t_block.weight", "proj.weight", "memory_proj.weight",
                         "align_masked_video.conv.weight", "embed_anchors.gate", "embed_anchors.proj.weight"]
                        + [f"transformer.{e}.projection.{w}.weight" for e in ("y_embedder", "t_embedder") for w in ("w1", "w2", "w3")])


def _f32_capable_sources(sd: Dict[str, torch.Tensor], cfg: SAMAudioConfig, device) -> Dict[str, "callable"]:
    """engine name -> thunk producing the fp32 operand of an F32-capable weight from the checkpoint"""
    def W(key):
        return sd[key].detach().to(device=device, dtype=torch.float32)

    P, c2 = "transformer.", cfg.transformer.out_channels
    src = {"w_out": lambda: W(P + "output.weight"), "tb_w": lambda: W(P + "t_block.weight"),
           "proj_wy": lambda: W("proj.weight")[:, :c2], "proj_wf": lambda: W("proj.weight")[:, 2 * c2:],
           "mem_w": lambda: W("memory_proj.weight"), "vid_w": lambda: W("align_masked_video.conv.weight").squeeze(-1),
           "anc_w": lambda: torch.tanh(W("embed_anchors.gate")).reshape(1, 1) * W("embed_anchors.proj.weight")}
    for pre, name in (("y", "y_embedder"), ("t", "t_embedder")):
        Q = f"{P}{name}.projection."
        src[f"{pre}_w13"] = lambda Q=Q: _interleave16(W(Q + "w1.weight"), W(Q + "w3.weight"))
        src[f"{pre}_w2"] = lambda Q=Q: W(Q + "w2.weight")
    return src


def convert_dit_f32(sd: Dict[str, torch.Tensor], cfg: SAMAudioConfig, device, classes: int) -> Dict[str, torch.Tensor]:
    """Only the "<name>.f32" operand copies of the F32-capable classes in the mask `classes` (hip.CLS bits)."""
    from . import hip
    return {name + ".f32": make().contiguous() for name, make in _f32_capable_sources(sd, cfg, device).items()
            if classes & hip.CLS[_F32_WEIGHT_CLASS[name]]}


def convert_dit_x3(tensors: Dict[str, torch.Tensor], n_layers: int, half: torch.dtype, classes: int,
                   ktm: bool = True) -> Dict[str, torch.Tensor]:
    """The "<name>.x3" split weights of the classes in the mask `classes` (hip.CLS bits of hip.X3_WEIGHTS), made from the fp32
    engine tensors `tensors` (convert_dit's output for act_dtype float32: the row permutations and interleaves are already in)."""
    from . import hip
    out: Dict[str, torch.Tensor] = {}
    for i in range(n_layers):
        for leaf, cls in hip.X3_WEIGHTS.items():
            if classes & hip.CLS[cls]:
                out[f"L{i}.{leaf}.x3"] = x3_weight(tensors[f"L{i}.{leaf}"], half, ktm)
    if classes & hip.CLS["patch"]:   # [D, 3 taps x D] -> each tap's D columns split: [D, 3 taps x 3D]
        for n in (1, 2):
            w = tensors[f"patch{n}.w"]
            d = w.shape[0]
            w3 = torch.cat([x3_weight(w[:, j * d:(j + 1) * d], half, ktm=False) for j in range(3)], dim=1).contiguous()
            out[f"patch{n}.w.x3"] = ktm_layout(w3) if ktm else w3
    if classes & hip.CLS["ckv"]:
        out["c_wkv_all.x3"] = x3_weight(tensors["c_wkv_all"], half, ktm)
    return out


def convert_dit(sd: Dict[str, torch.Tensor], cfg: SAMAudioConfig, act_dtype: torch.dtype,
                device, alt16_leaves=(), f32_classes: Optional[int] = None, ktm: bool = False) -> Dict[str, torch.Tensor]:
    """`ktm` (16-bit models): the weights of the five big GEMM classes of the layers (wqkv, wo, c_wq, w13, w2) are stored
    K-tile-major (ktm_layout).
    `alt16_leaves`: per-layer weight names (of "wqkv", "wo", "c_wq", "w13", "w2") whose GEMM class reads bfloat16
    operands in a mixed-precision model (hip.ALT16_WEIGHTS): converted from fp32 to bfloat16 instead of `act_dtype`.
    `f32_classes` (16-bit models): mask of the F32-capable classes whose weights also get an fp32 copy under
    "<name>.f32" (None = all five, 0.5 GB at large* dims; SAMAudio passes the classes it will run in fp32)."""
    from . import hip
    want32 = hip.CLS_F32_CAPABLE if f32_classes is None else f32_classes
    t = cfg.transformer
    D, H, F = t.dim, t.n_heads, t.ffn_hidden
    out: Dict[str, torch.Tensor] = {}

    def f32(x):
        return x.detach().to(device=device, dtype=torch.float32).contiguous()

    ktm = bool(ktm) and act_dtype != torch.float32

    def op(x, leaf=None):  # GEMM operand
        dt = torch.bfloat16 if leaf in alt16_leaves else act_dtype
        w = x.detach().to(device=device, dtype=torch.float32).to(dt).contiguous()
        return ktm_layout(w) if (ktm and leaf in KTM_LEAVES) else w

    def op_f32(name, x):
        """GEMM operand of a class that may run in exact fp32 inside a 16-bit engine (samaudio.h SAMAUDIO_OPT_F32_CLASSES):
        the 16-bit copy under `name`, the fp32 one under `name + ".f32"` when its class is in `f32_classes`."""
        out[name] = op(x)
        if act_dtype != torch.float32 and want32 & hip.CLS[_F32_WEIGHT_CLASS[name]]:
            out[name + ".f32"] = f32(x)

    def W(key):
        return sd[key].detach().to(device=device, dtype=torch.float32)

    P = "transformer."
    for i in range(t.n_layers):
        L, E = f"{P}layers.{i}.", f"L{i}."
        out[E + "attn_norm"] = f32(sd[L + "attention_norm.weight"])
        out[E + "ffn_norm"] = f32(sd[L + "ffn_norm.weight"])
        out[E + "mod_table"] = f32(sd[L + "scale_shift_table"])
        out[E + "q_norm"] = f32(sd[L + "attention.q_norm.weight"])
        out[E + "k_norm"] = f32(sd[L + "attention.k_norm.weight"])
        out[E + "c_q_norm"] = f32(sd[L + "cross_attention.q_norm.weight"])
        out[E + "wqkv"] = op(torch.cat([_head_major(W(L + f"attention.{n}.weight"), H) for n in ("wq", "wk", "wv")]), "wqkv")
        out[E + "wo"] = op(W(L + "attention.wo.weight"), "wo")
        out[E + "c_wq"] = op(_head_major(W(L + "cross_attention.wq.weight"), H), "c_wq")
        out[E + "c_wo"] = op(W(L + "cross_attention.wo.weight"))
        out[E + "w13"] = op(_interleave16(W(L + "feed_forward.w1.weight"), W(L + "feed_forward.w3.weight")), "w13")
        out[E + "w2"] = op(W(L + "feed_forward.w2.weight"), "w2")
    # cross-attention K|V projections and k-norm weights of all layers, stacked: one GEMM per evaluation
    out["c_wkv_all"] = op(torch.cat([_head_major(W(f"{P}layers.{i}.cross_attention.{n}.weight"), H)
                                     for i in range(t.n_layers) for n in ("wk", "wv")]))
    out["c_k_norm_all"] = f32(torch.stack([sd[f"{P}layers.{i}.cross_attention.k_norm.weight"] for i in range(t.n_layers)]))
    out["final_table"] = f32(sd[P + "final_layer_scale_shift_table"])
    out["final_norm"] = f32(sd[P + "norm.weight"])
    op_f32("w_out", W(P + "output.weight"))
    for n, blk in ((1, "block1"), (2, "block2")):
        B = f"{P}x_embedder.block.{blk}."
        out[f"patch{n}.gn_w"] = f32(sd[B + "groupnorm.weight"])
        out[f"patch{n}.gn_b"] = f32(sd[B + "groupnorm.bias"])
        out[f"patch{n}.w"] = op(W(B + "project.weight").permute(0, 2, 1).reshape(D, 3 * D))
        out[f"patch{n}.b"] = f32(sd[B + "project.bias"])
    for pre, name in (("y", "y_embedder"), ("t", "t_embedder")):
        Q = f"{P}{name}.projection."
        op_f32(f"{pre}_w13", _interleave16(W(Q + "w1.weight"), W(Q + "w3.weight")))
        op_f32(f"{pre}_w2", W(Q + "w2.weight"))
    op_f32("tb_w", W(P + "t_block.weight"))
    out["tb_b"] = f32(sd[P + "t_block.bias"])

    c2 = t.out_channels
    proj = W("proj.weight")
    assert proj.shape[1] == 3 * c2, "proj expects [noisy | zeros | features]"
    op_f32("proj_wy", proj[:, :c2])
    op_f32("proj_wf", proj[:, 2 * c2:])
    out["proj_b"] = f32(sd["proj.bias"])
    op_f32("mem_w", W("memory_proj.weight"))
    out["mem_b"] = f32(sd["memory_proj.bias"])
    op_f32("vid_w", W("align_masked_video.conv.weight").squeeze(-1))
    out["vid_b"] = f32(sd["align_masked_video.conv.bias"])
    out["vid_ln_w"] = f32(sd["align_masked_video.layer_norm.weight"])
    out["vid_ln_b"] = f32(sd["align_masked_video.layer_norm.bias"])
    out["vid_gate"] = f32(sd["align_masked_video.gate"].reshape(1))
    out["anc_emb"] = f32(sd["embed_anchors.embed.weight"])
    op_f32("anc_w", torch.tanh(W("embed_anchors.gate")).reshape(1, 1) * W("embed_anchors.proj.weight"))

    # fp32 tables, computed with the reference's op sequence on the CPU
    hd = t.head_dim
    freqs = 1.0 / (t.rope_theta ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd))
    ang = torch.outer(torch.arange(t.max_positions), freqs).float()
    out["rope_cos"], out["rope_sin"] = f32(ang.cos()), f32(ang.sin())
    half = t.frequency_embedding_dim // 2
    out["t_freqs"] = f32(torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half))
    half = D // 2
    out["mem_inv_freq"] = f32(torch.exp(-math.log(10000) * torch.arange(half).float() / half))
    return out


def convert_codec(sd: Dict[str, torch.Tensor], cfg: SAMAudioConfig, act_dtype: torch.dtype, device,
                  prefix: str = "audio_codec.", with_decoder: bool = True) -> Dict[str, torch.Tensor]:
    """`cfg` needs an `.audio_codec` DACVAEConfig.  with_decoder=False converts the encoder + quantizer.in_proj only
    (the Judge's DACVAEEncoder, reference codec.py:42-78)."""
    c = cfg.audio_codec
    slab = 64 if act_dtype != torch.float32 else 32
    out: Dict[str, torch.Tensor] = {}

    def f32(x):
        return x.detach().to(device=device, dtype=torch.float32).contiguous()

    def op(x):
        return x.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()

    def conv(name):  # [Cout, Cin, k] -> [Cout, k*Cin]
        w = _conv_weight(sd, prefix + name).to(device)
        return op(_pad_k(w.permute(0, 2, 1).reshape(w.shape[0], -1), slab))

    def alpha(name):
        return f32(sd[prefix + name + ".alpha"].reshape(-1))

    def bias(name):
        return f32(sd[prefix + name + ".bias"])

    def res_unit(src: str, dst: str):
        out[dst + "a1"], out[dst + "w1"], out[dst + "b1"] = alpha(src + ".block.0"), conv(src + ".block.1"), bias(src + ".block.1")
        out[dst + "a2"], out[dst + "w2"], out[dst + "b2"] = alpha(src + ".block.2"), conv(src + ".block.3"), bias(src + ".block.3")

    # encoder ------------------------------------------------------------------------------------
    w_in = _conv_weight(sd, prefix + "encoder.block.0").to(device)  # [C0, 1, 7]
    w8 = w_in.new_zeros(w_in.shape[0], 8, 8)                          # [C0, tap(8), channel(8)]
    w8[:, :7, 0] = w_in[:, 0, :]
    out["enc.in.w"], out["enc.in.b"] = op(w8.reshape(w_in.shape[0], 64)), bias("encoder.block.0")
    for i in range(len(c.encoder_rates)):
        B = f"encoder.block.{i + 1}.block."
        for j in range(3):
            res_unit(f"{B}{j}", f"enc.s{i}.r{j}.")
        out[f"enc.s{i}.a"] = alpha(B + "3")
        out[f"enc.s{i}.down.w"], out[f"enc.s{i}.down.b"] = conv(B + "4"), bias(B + "4")
    out["enc.out.a"] = alpha("encoder.block.5")
    out["enc.out.w"], out["enc.out.b"] = conv("encoder.block.6"), bias("encoder.block.6")
    w_ip = _conv_weight(sd, prefix + "quantizer.in_proj").to(device).squeeze(-1)  # [2*cd, latent]
    out["enc.proj.w"] = op(w_ip[: c.codebook_dim])                                   # mean half only (codec.py:68)
    out["enc.proj.b"] = f32(sd[prefix + "quantizer.in_proj.bias"][: c.codebook_dim])
    if not with_decoder:
        return out
    # decoder ------------------------------------------------------------------------------------
    out["dec.proj.w"] = op(_conv_weight(sd, prefix + "quantizer.out_proj").to(device).squeeze(-1))
    out["dec.proj.b"] = bias("quantizer.out_proj")
    out["dec.in.w"], out["dec.in.b"] = conv("decoder.model.0"), bias("decoder.model.0")
    for i, s in enumerate(c.decoder_rates):
        B = f"decoder.model.{i + 1}.block."
        out[f"dec.s{i}.a"] = alpha(B + "0")
        w = _conv_weight(sd, prefix + B + "1", transposed=True).to(device)  # ConvTranspose1d [Cin, Cout, 2s]
        cin, cout, k = w.shape
        assert k == 2 * s
        wg = torch.stack([w[:, :, s:], w[:, :, :s]], dim=0)            # [j, Cin, Cout, r]: j=0 <- x[q-1], j=1 <- x[q]
        out[f"dec.s{i}.up.w"] = op(wg.permute(3, 2, 0, 1).reshape(s * cout, 2 * cin))
        out[f"dec.s{i}.up.b"] = bias(B + "1")
        for j in range(3):
            res_unit(f"{B}{j + 2}", f"dec.s{i}.r{j}.")
    out["dec.out.a"] = alpha("decoder.model.5")
    out["dec.out.w"], out["dec.out.b"] = conv("decoder.model.6"), bias("decoder.model.6")
    return out


def convert_codec_x3(tensors: Dict[str, torch.Tensor], half: torch.dtype, min_out: int = 256) -> Dict[str, torch.Tensor]:
    """The "<name>.x3" twins of the codec convolutions with >= `min_out` output channels (include/samaudio.h SAMAUDIO_OPT_X3_CLASSES,
    class CODEC), made from the fp32 engine tensors of convert_codec: a weight row [K] = [block][Cin] becomes [block][W_hi | W_lo |
    W_hi], shape [N, K / Cin, 3 Cin] - Cin = the channel count of one input row (a tap of a dilated convolution, a time step of a
    strided / transposed one), so that the activation buffer split row by row into [lo | hi | hi] lines up with it."""
    out: Dict[str, torch.Tensor] = {}
    for name, w in tensors.items():
        if w.dim() != 2 or w.dtype != torch.float32 or not (name.startswith("enc.") or name.startswith("dec.")) or name.endswith(".x3"):
            continue
        n, k = w.shape
        leaf = name.rsplit(".", 2)[-2:] if name.count(".") >= 2 else [name]
        if name.endswith((".w1", ".w2")):
            cin = n
        elif name.endswith("down.w"):
            cin = n // 2
        elif name == "enc.out.w":
            cin = k // 3
        elif name == "dec.proj.w":
            cin = k
        elif name == "dec.in.w":
            cin = k // 7
        elif name.endswith("up.w"):
            cin = k // 2
        else:
            continue
        if n < min_out or cin % 8 or k % cin or (3 * k) % 64:
            continue
        blocks = w.reshape(n, k // cin, cin)
        hi = blocks.clamp(-65504.0, 65504.0).to(half) if half == torch.float16 else blocks.to(half)
        lo = (blocks - hi.float()).to(half)
        out[name + ".x3"] = torch.cat([hi, lo, hi], dim=2).contiguous()
    return out


def fly16_weight(w: torch.Tensor, half: torch.dtype) -> torch.Tensor:
    """An fp32 GEMM weight [N, K] (K % 32 == 0) in the split layout of GemmParams.flags bit 14 (csrc/common.h GEMM_FLAG_W_FLY16):
    per row and 32-k slab, 128 bytes = 4 chunks of hi halves + 4 chunks of lo halves, chunk c = k in {4c .. 4c+3, 16+4c .. 16+4c+3} -
    one lane's operand of a 16x16x32 MFMA.  hi = rn16(clamp(w)), lo = rn16(w - hi): the bits gemm.hip's register split computes.
    Returned as a 16-bit tensor [N, 2K] (the byte size and row stride of the fp32 matrix)."""
    n, k = w.shape
    assert w.dtype == torch.float32 and k % 32 == 0
    hi = w.clamp(-65504.0, 65504.0).to(half) if half == torch.float16 else w.to(half)
    lo = (w - hi.float()).to(half)
    # k = 16 j + 4 c + e  ->  [slab][c][j][e]
    parts = [t.reshape(n, k // 32, 2, 4, 4).permute(0, 1, 3, 2, 4).reshape(n, k // 32, 32) for t in (hi, lo)]
    return torch.cat(parts, dim=2).reshape(n, 2 * k).contiguous()


def fly16_to_f32(t: torch.Tensor) -> torch.Tensor:
    """Inverse of fly16_weight up to the split (hi + lo as fp32): tests."""
    n, k2 = t.shape
    k = k2 // 2
    b = t.reshape(n, k // 32, 2, 4, 2, 4).float()          # [slab][hi|lo][c][j][e]
    v = b[:, :, 0] + b[:, :, 1]
    return v.permute(0, 1, 3, 2, 4).reshape(n, k)


def convert_codec_fly16(tensors: Dict[str, torch.Tensor], half: torch.dtype, max_out: int = 256) -> Dict[str, torch.Tensor]:
    """The "<name>.fly" twins of the codec convolutions with < `max_out` output channels - the ones convert_codec_x3 leaves to the
    fp32 kernel with operands split on the fly (SAMAUDIO_OPT_X3_CLASSES bit CODEC): their weights split ONCE, in the layout that
    kernel's fragment reads want (fly16_weight), so that its K loop splits activations only."""
    out: Dict[str, torch.Tensor] = {}
    for name, w in tensors.items():
        if w.dim() != 2 or w.dtype != torch.float32 or not (name.startswith("enc.") or name.startswith("dec.")) or name.endswith((".x3", ".fly")):
            continue
        n, k = w.shape
        if n >= max_out or k % 32 or not name.endswith(".w") and not name.endswith((".w1", ".w2")):
            continue
        out[name + ".fly"] = fly16_weight(w, half)
    return out


def split_missing_unexpected(sd_keys, cfg: SAMAudioConfig) -> Tuple[List[str], List[str]]:
    """Key bookkeeping of reference SAMAudio.load_state_dict (model.py:346-359)."""
    want = set(expected_keys(cfg))
    have = set(sd_keys)

    def canon(k):  # weight-norm spellings count as the plain weight
        return re.sub(r"\.(weight_g|weight_v|parametrizations\.weight\.original[01])$", ".weight", k)

    have_c = {canon(k) for k in have}
    missing = sorted(k for k in want if k not in have_c and not OPTIONAL_KEY_RE.search(k))
    unexpected = sorted(k for k in have if canon(k) not in want and not OPTIONAL_KEY_RE.search(k)
                        and not IGNORED_KEY_RE.search(k))
    return missing, unexpected
