"""`T5EncoderHIP` - the T5 encoder stack behind `T5TextEncoder.forward` on the HIP library
(reference sam_audio/model/text_encoder.py:11-37:
`self.model = transformers.T5EncoderModel.from_pretrained("t5-base")`,
`self.model(input_ids=..., attention_mask=..., output_hidden_states=True)["last_hidden_state"]`).

Host code only: it maps the `shared.* / encoder.*` state_dict keys of `T5EncoderModel` onto the engine's tensors
(one-time re-layout, incl. the relative-position bias evaluated per signed distance), sizes the workspace and calls
`samaudio_t5_*`; every arithmetic step of the forward is a HIP kernel (sam_audio_amd/csrc/t5.hip).  Tokenisation stays
with the Hugging Face tokenizer exactly as in the reference.  No CPU / eager fallback.

T5EncoderModel key                                          engine tensor
  shared.weight [V, D] (= encoder.embed_tokens.weight)      emb [V, D] f32
  encoder.block.0.layer.0.SelfAttention
      .relative_attention_bias.weight [buckets, H]          rel_bias [H, 2*max_len - 1] f32: column (k - q) + max_len - 1
  encoder.block.{i}.layer.0.layer_norm.weight               L{i}.ln1
  encoder.block.{i}.layer.0.SelfAttention.{q,k,v}.weight    L{i}.wqkv [3 * H * d_kv, D]   (rows q | k | v)
  encoder.block.{i}.layer.0.SelfAttention.o.weight          L{i}.wo [D, H * d_kv]
  encoder.block.{i}.layer.1.layer_norm.weight               L{i}.ln2
  encoder.block.{i}.layer.1.DenseReluDense.wi / wo          L{i}.wi [F, D], L{i}.wo2 [D, F]
  encoder.final_layer_norm.weight                           final_ln
"""
from __future__ import annotations

import ctypes as C
import math
import re
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch

from . import hip
from .judge import _ensure_ws, _register

ACTS = {"relu": hip.ACT_RELU, "gelu_new": hip.ACT_GELU_TANH}


@dataclass
class T5Dims:
    """The fields of `transformers.T5Config` the encoder stack depends on (defaults = t5-base)."""
    vocab_size: int = 32128
    d_model: int = 768
    d_kv: int = 64
    num_heads: int = 12
    d_ff: int = 3072
    num_layers: int = 12
    relative_attention_num_buckets: int = 32
    relative_attention_max_distance: int = 128
    dense_act_fn: str = "relu"
    is_gated_act: bool = False
    layer_norm_epsilon: float = 1e-6
    max_len: int = 512                 # longest sequence the engine's relative-position table covers

    @classmethod
    def from_hf(cls, cfg: Any, max_len: Optional[int] = 512) -> "T5Dims":
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        ffp = get("feed_forward_proj", "relu")
        gated = bool(get("is_gated_act", str(ffp).startswith("gated-")))
        act = get("dense_act_fn", str(ffp).split("-")[-1])
        if act == "gelu" and gated:     # transformers' backwards-compatibility rule for "gated-gelu"
            act = "gelu_new"
        return cls(vocab_size=get("vocab_size"), d_model=get("d_model"), d_kv=get("d_kv"), num_heads=get("num_heads"),
                   d_ff=get("d_ff"), num_layers=get("num_layers"),
                   relative_attention_num_buckets=get("relative_attention_num_buckets", 32),
                   relative_attention_max_distance=get("relative_attention_max_distance", 128),
                   dense_act_fn=act, is_gated_act=gated, layer_norm_epsilon=get("layer_norm_epsilon", 1e-6),
                   max_len=min(int(max_len or 512), 512))

    def check_supported(self) -> None:
        if self.is_gated_act or self.dense_act_fn not in ACTS:
            raise NotImplementedError(
                f"T5 feed-forward '{'gated-' if self.is_gated_act else ''}{self.dense_act_fn}' is not built on the HIP "
                "library (t5-base, the reference's text encoder, is non-gated ReLU)")


def relative_position_bucket(relative_position: torch.Tensor, num_buckets: int, max_distance: int) -> torch.Tensor:
    """The bidirectional bucket rule of T5Attention._relative_position_bucket, with the same fp32 torch ops (the
    buckets of the distances 16, 32, 64 sit on a float rounding edge, so the formula is not re-derived)."""
    num_buckets //= 2
    buckets = (relative_position > 0).to(torch.long) * num_buckets
    rel = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = rel < max_exact
    large = max_exact + (torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact)
                         * (num_buckets - max_exact)).to(torch.long)
    large = torch.min(large, torch.full_like(large, num_buckets - 1))
    return buckets + torch.where(is_small, rel, large)


def relative_bias_table(weight: torch.Tensor, dims: T5Dims) -> torch.Tensor:
    """[H, 2*max_len - 1] f32: entry (h, d + max_len - 1) = relative_attention_bias[bucket(d)][h] for the signed
    distance d = key position - query position (what T5Attention.compute_bias gathers per call)."""
    d = torch.arange(-(dims.max_len - 1), dims.max_len, dtype=torch.long)
    b = relative_position_bucket(d, dims.relative_attention_num_buckets, dims.relative_attention_max_distance)
    return weight.detach().float().cpu()[b].t().contiguous()


def expected_keys(dims: T5Dims) -> List[str]:
    keys = ["shared.weight", "encoder.final_layer_norm.weight",
            "encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"]
    for i in range(dims.num_layers):
        p = f"encoder.block.{i}.layer."
        keys += [p + f"0.SelfAttention.{n}.weight" for n in "qkvo"]
        keys += [p + "0.layer_norm.weight", p + "1.layer_norm.weight", p + "1.DenseReluDense.wi.weight",
                 p + "1.DenseReluDense.wo.weight"]
    return keys


def convert_t5(sd: Dict[str, torch.Tensor], dims: T5Dims, act_dtype: torch.dtype, device) -> Dict[str, torch.Tensor]:
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
    act = lambda t: t.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()  # noqa: E731
    out: Dict[str, torch.Tensor] = {"emb": f32(sd["shared.weight"]), "final_ln": f32(sd["encoder.final_layer_norm.weight"])}
    out["rel_bias"] = relative_bias_table(
        sd["encoder.block.0.layer.0.SelfAttention.relative_attention_bias.weight"], dims).to(device)
    for i in range(dims.num_layers):
        s, d = f"encoder.block.{i}.layer.", f"L{i}."
        out[d + "ln1"], out[d + "ln2"] = f32(sd[s + "0.layer_norm.weight"]), f32(sd[s + "1.layer_norm.weight"])
        out[d + "wqkv"] = act(torch.cat([sd[s + f"0.SelfAttention.{n}.weight"].detach().float() for n in "qkv"], dim=0))
        out[d + "wo"] = act(sd[s + "0.SelfAttention.o.weight"])
        out[d + "wi"], out[d + "wo2"] = act(sd[s + "1.DenseReluDense.wi.weight"]), act(sd[s + "1.DenseReluDense.wo.weight"])
    return out


class T5EncoderHIP:
    """`enc(input_ids [B, Lt] int64, attention_mask [B, Lt]) -> last_hidden_state [B, Lt, d_model] float32`."""

    def __init__(self, dims: T5Dims, precision: str = "fp32", device: Optional[str] = None):
        hip.check_precision(precision)
        dims.check_supported()
        self.dims = dims
        self.precision = precision
        self.device = torch.device(device) if device is not None else None
        self._lib = hip.lib(hip.operands_for(precision))
        self._h = C.c_void_p()
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self._loaded = False
        tc = hip.T5Config(precision=hip.precision_code(precision), vocab=dims.vocab_size, d_model=dims.d_model,
                          d_kv=dims.d_kv, heads=dims.num_heads, d_ff=dims.d_ff, layers=dims.num_layers,
                          max_len=dims.max_len, act=ACTS[dims.dense_act_fn], ln_eps=dims.layer_norm_epsilon)
        hip.check(self._lib.samaudio_t5_create(C.byref(tc), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.samaudio_t5_destroy(self._h)
            self._h = None

    @property
    def act_dtype(self) -> torch.dtype:
        return hip.act_dtype(self.precision)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """`T5EncoderModel.state_dict()` keys (optionally below `model.` / `text_encoder.model.`);
        `encoder.embed_tokens.weight` is the tied copy of `shared.weight` and is ignored."""
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hip.require_gpu(self.device, "T5EncoderHIP")
        pref = re.compile(r"^(text_encoder\.)?(model\.)?(?=shared\.|encoder\.)")
        sd = {pref.sub("", k): v for k, v in state_dict.items()}
        sd.pop("encoder.embed_tokens.weight", None)
        want = set(expected_keys(self.dims))
        missing, unexpected = sorted(want - set(sd)), sorted(set(sd) - want)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Missing keys: {missing}, unexpected_keys: {unexpected}")
        if not missing:
            with torch.cuda.device(self.device):
                _register(self._lib.samaudio_t5_set_tensor, self._h, self._tensors,
                          convert_t5(sd, self.dims, self.act_dtype, self.device))
                hip.check(self._lib.samaudio_t5_finalize(self._h))
            self._loaded = True
        return missing, unexpected

    @torch.inference_mode()
    def encode(self, input_ids: torch.Tensor, attention_mask: torch.Tensor, ids_checked: bool = False) -> torch.Tensor:
        """`ids_checked`: the caller has validated the token ids on the host (T5TextEncoder does, on the tokenizer's CPU output).
        Ids that arrive on the CPU are checked here for free; ids that arrive on the GPU unchecked cost a blocking device -> host
        read (it waits for everything queued on the stream - inside separate() that is the DAC encode)."""
        if not self._loaded:
            raise hip.SamAudioHipError("T5EncoderHIP: no weights loaded")
        assert input_ids.dim() == 2 and attention_mask.shape == input_ids.shape, "input_ids / attention_mask must be [B, Lt]"
        rows, tokens = input_ids.shape
        if tokens > self.dims.max_len:
            raise ValueError(f"{tokens} tokens exceed the encoder's max_len {self.dims.max_len}")
        if rows and tokens and not (ids_checked and input_ids.device.type != "cpu"):
            ids_host = input_ids.detach().to("cpu", torch.int64)
            if int(ids_host.min()) < 0 or int(ids_host.max()) >= self.dims.vocab_size:
                raise IndexError(f"token id outside [0, {self.dims.vocab_size})")   # nn.Embedding raises IndexError too
        with torch.cuda.device(self.device):
            ids = input_ids.to(self.device, torch.int64).contiguous()
            mask = (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous()
            out = torch.empty(rows, tokens, self.dims.d_model, device=self.device, dtype=torch.float32)
            if rows == 0 or tokens == 0:
                return out
            need = self._lib.samaudio_t5_workspace_bytes(self._h, rows, tokens)
            _ensure_ws(self, need, lambda p, b: self._lib.samaudio_t5_set_workspace(self._h, p, b))
            hip.check(self._lib.samaudio_t5_encode(self._h, hip.ptr(ids), hip.ptr(mask), rows, tokens, hip.ptr(out),
                                                   hip.current_stream_ptr()))
        return out

    __call__ = encode


def encoder_flops(dims: T5Dims, rows: int, tokens: int) -> float:
    """Algorithmic FLOPs (2 x MACs) of one encode: per layer q|k|v + scores + PV + o + the two feed-forward GEMMs."""
    m, inner = rows * tokens, dims.num_heads * dims.d_kv
    per = 2.0 * m * dims.d_model * 3 * inner + 4.0 * rows * tokens * tokens * inner + 2.0 * m * inner * dims.d_model
    per += 4.0 * m * dims.d_model * dims.d_ff
    return per * dims.num_layers
