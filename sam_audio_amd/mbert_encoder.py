"""`ModernBertHIP` - the ModernBERT text tower of the Judge reranker and the PE-A-Frame span predictor on the HIP library
(reference sam_audio/model/judge.py:48 `AutoModel.from_config(ModernBertConfig(...))`,
:74-88 `self.text_model(input_ids=..., attention_mask=..., output_hidden_states=True).hidden_states[nth_text_layer]`).

Host code only: it maps the `ModernBertModel.state_dict()` keys onto the engine's tensors (one-time re-layout, incl. the two
rotary tables), sizes the workspace and calls `samaudio_mbert_*`; every arithmetic step is a HIP kernel
(sam_audio_amd/csrc/mbert.hip).  Tokenisation stays with the Hugging Face tokenizer as in the reference.  No CPU fallback.

ModernBertModel key                               engine tensor
  embeddings.tok_embeddings.weight [V, D]           emb [V, D] f32
  embeddings.norm.weight                            emb_ln
  layers.{i}.attn_norm.weight (i > 0)               L{i}.ln1            (layer 0: nn.Identity)
  layers.{i}.attn.Wqkv.weight [3D, D]               L{i}.wqkv           (rows q | k | v, heads contiguous inside each)
  layers.{i}.attn.Wo.weight                         L{i}.wo
  layers.{i}.mlp_norm.weight                        L{i}.ln2
  layers.{i}.mlp.Wi.weight [2F, D]                  L{i}.wi             (rows input | gate)
  layers.{i}.mlp.Wo.weight [D, F]                   L{i}.wo2
  final_norm.weight                                 final_ln
  (rotary embedding has no parameters)              rope_cos / rope_sin [2, max_len, hd]: [0] global layers, [1] local layers
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch

from . import hip
from .judge_util import ensure_ws, register


@dataclass
class MBertDims:
    """The fields of `transformers.ModernBertConfig` the encoder depends on (defaults = ModernBERT-base)."""
    vocab_size: int = 50368
    hidden_size: int = 768
    num_attention_heads: int = 12
    intermediate_size: int = 1152
    num_hidden_layers: int = 22
    global_attn_every_n_layers: int = 3
    local_attention: int = 128
    global_rope_theta: float = 160000.0
    local_rope_theta: float = 10000.0
    norm_eps: float = 1e-5
    hidden_activation: str = "gelu"
    norm_bias: bool = False
    attention_bias: bool = False
    mlp_bias: bool = False
    max_len: int = 512

    @classmethod
    def from_hf(cls, cfg: Any, max_len: int = 512) -> "MBertDims":
        get = (lambda k, d=None: cfg.get(k, d)) if isinstance(cfg, dict) else (lambda k, d=None: getattr(cfg, k, d))
        types = get("layer_types")
        every = get("global_attn_every_n_layers", 3)
        if types:   # transformers >= 5 carries the per-layer list; it must be the regular pattern the engine implements
            n = next((i for i in range(1, len(types)) if types[i] == "full_attention"), len(types))
            every = n
            if any((t == "full_attention") != (i % every == 0) for i, t in enumerate(types)):
                raise NotImplementedError(f"irregular ModernBERT layer_types {types}")
        rp = get("rope_parameters") or {}
        g_theta = (rp.get("full_attention") or {}).get("rope_theta", get("global_rope_theta", 160000.0))
        l_theta = (rp.get("sliding_attention") or {}).get("rope_theta", get("local_rope_theta", 10000.0))
        return cls(vocab_size=get("vocab_size"), hidden_size=get("hidden_size"), num_attention_heads=get("num_attention_heads"),
                   intermediate_size=get("intermediate_size"), num_hidden_layers=get("num_hidden_layers"),
                   global_attn_every_n_layers=every, local_attention=get("local_attention", 128),
                   global_rope_theta=float(g_theta), local_rope_theta=float(l_theta), norm_eps=get("norm_eps", 1e-5),
                   hidden_activation=get("hidden_activation", "gelu"), norm_bias=bool(get("norm_bias", False)),
                   attention_bias=bool(get("attention_bias", False)), mlp_bias=bool(get("mlp_bias", False)),
                   max_len=min(int(max_len), 512))

    def check_supported(self) -> None:
        if self.norm_bias or self.attention_bias or self.mlp_bias or self.hidden_activation != "gelu":
            raise NotImplementedError("ModernBERT with biases / an activation other than gelu is not built on the HIP "
                                      "library")


def rope_tables(dims: MBertDims):
    """cos / sin [2, max_len, head_dim] as ModernBertRotaryEmbedding forms them (fp32; emb = cat(freqs, freqs))."""
    hd = dims.hidden_size // dims.num_attention_heads
    pos = torch.arange(dims.max_len, dtype=torch.float32)
    out = []
    for theta in (dims.global_rope_theta, dims.local_rope_theta):
        inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
        freqs = (inv[None, :, None].float() @ pos[None, None, :]).transpose(1, 2)[0]      # [max_len, hd / 2], HF's matmul
        emb = torch.cat((freqs, freqs), dim=-1)
        out.append((emb.cos(), emb.sin()))
    return torch.stack([o[0] for o in out]).contiguous(), torch.stack([o[1] for o in out]).contiguous()


def expected_keys(dims: MBertDims) -> List[str]:
    keys = ["embeddings.tok_embeddings.weight", "embeddings.norm.weight", "final_norm.weight"]
    for i in range(dims.num_hidden_layers):
        p = f"layers.{i}."
        keys += [p + "attn.Wqkv.weight", p + "attn.Wo.weight", p + "mlp_norm.weight", p + "mlp.Wi.weight", p + "mlp.Wo.weight"]
        if i > 0:
            keys.append(p + "attn_norm.weight")
    return keys


def convert_mbert(sd: Dict[str, torch.Tensor], dims: MBertDims, act_dtype: torch.dtype, device) -> Dict[str, torch.Tensor]:
    f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
    act = lambda t: t.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()  # noqa: E731
    cos, sin = rope_tables(dims)
    out: Dict[str, torch.Tensor] = {
        "emb": f32(sd["embeddings.tok_embeddings.weight"]), "emb_ln": f32(sd["embeddings.norm.weight"]),
        "final_ln": f32(sd["final_norm.weight"]), "zeros": torch.zeros(dims.hidden_size, dtype=torch.float32, device=device),
        "rope_cos": cos.to(device), "rope_sin": sin.to(device)}
    for i in range(dims.num_hidden_layers):
        s, d = f"layers.{i}.", f"L{i}."
        if i > 0:
            out[d + "ln1"] = f32(sd[s + "attn_norm.weight"])
        out[d + "wqkv"], out[d + "wo"] = act(sd[s + "attn.Wqkv.weight"]), act(sd[s + "attn.Wo.weight"])
        out[d + "ln2"] = f32(sd[s + "mlp_norm.weight"])
        out[d + "wi"], out[d + "wo2"] = act(sd[s + "mlp.Wi.weight"]), act(sd[s + "mlp.Wo.weight"])
    return out


class ModernBertHIP:
    """`tower(input_ids [B, Lt], attention_mask [B, Lt] | None, nth=None) -> [B, Lt, hidden] float32`:
    nth None = last_hidden_state, otherwise transformers' `hidden_states[nth]`."""

    def __init__(self, dims: MBertDims, precision: str = "fp32", device: Optional[str] = None):
        hip.check_precision(precision)
        dims.check_supported()
        self.dims, self.precision = dims, precision
        self.device = torch.device(device) if device is not None else None
        self._lib = hip.lib(hip.operands_for(precision))
        self._h = C.c_void_p()
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self._loaded = False
        mc = hip.MBertConfig(precision=hip.precision_code(precision), vocab=dims.vocab_size, hidden=dims.hidden_size,
                             heads=dims.num_attention_heads, intermediate=dims.intermediate_size,
                             layers=dims.num_hidden_layers, global_every=dims.global_attn_every_n_layers,
                             window=dims.local_attention // 2, max_len=dims.max_len, ln_eps=dims.norm_eps)
        hip.check(self._lib.samaudio_mbert_create(C.byref(mc), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.samaudio_mbert_destroy(self._h)
            self._h = None

    @classmethod
    def from_module(cls, module, device, precision: str = "fp32") -> "ModernBertHIP":
        """Build from a `transformers.ModernBertModel` (its config and weights); the module itself is not used afterwards."""
        tower = cls(MBertDims.from_hf(module.config), precision=precision, device=str(device))
        tower.load_state_dict(module.state_dict())
        return tower

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hip.require_gpu(self.device, "ModernBertHIP")
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()}
        want = set(expected_keys(self.dims))
        missing, unexpected = sorted(want - set(sd)), sorted(set(sd) - want)
        if strict and (missing or unexpected):
            raise RuntimeError(f"Missing keys: {missing}, unexpected_keys: {unexpected}")
        if not missing:
            with torch.cuda.device(self.device):
                register(self._lib.samaudio_mbert_set_tensor, self._h, self._tensors,
                         convert_mbert(sd, self.dims, hip.act_dtype(self.precision), self.device))
                hip.check(self._lib.samaudio_mbert_finalize(self._h))
            self._loaded = True
        return missing, unexpected

    @torch.inference_mode()
    def encode(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
               nth: Optional[int] = None, last_prenorm: bool = True) -> torch.Tensor:
        """nth None = last_hidden_state; 0 <= nth < layers = `hidden_states[nth]`; nth == layers = `hidden_states[layers]` as
        transformers 4.48 - 4.5x define it (the last layer's output BEFORE final_norm; `last_prenorm=True`, the default) or as
        transformers 5.x define it (= last_hidden_state; `last_prenorm=False`)."""
        if not self._loaded:
            raise hip.SamAudioHipError("ModernBertHIP: no weights loaded")
        assert input_ids.dim() == 2, "input_ids must be [B, Lt]"
        rows, tokens = input_ids.shape
        if tokens > self.dims.max_len:
            raise ValueError(f"{tokens} tokens exceed the text tower's max_len {self.dims.max_len}")
        if nth is not None and not 0 <= nth <= self.dims.num_hidden_layers:
            raise IndexError(f"hidden_states[{nth}] of a {self.dims.num_hidden_layers}-layer tower")
        ids_host = input_ids.detach().to("cpu", torch.int64)
        if rows and tokens and (int(ids_host.min()) < 0 or int(ids_host.max()) >= self.dims.vocab_size):
            raise IndexError(f"token id outside [0, {self.dims.vocab_size})")
        with torch.cuda.device(self.device):
            ids = input_ids.to(self.device, torch.int64).contiguous()
            mask = (torch.ones_like(ids, dtype=torch.uint8) if attention_mask is None
                    else (attention_mask.to(self.device) != 0).to(torch.uint8).contiguous())
            out = torch.empty(rows, tokens, self.dims.hidden_size, device=self.device, dtype=torch.float32)
            if rows == 0 or tokens == 0:
                return out
            need = self._lib.samaudio_mbert_workspace_bytes(self._h, rows, tokens)
            ensure_ws(self, need, lambda p, b: self._lib.samaudio_mbert_set_workspace(self._h, p, b))
            hip.check(self._lib.samaudio_mbert_encode(self._h, hip.ptr(ids), hip.ptr(mask), rows, tokens,
                                                      -1 if nth is None or (nth == self.dims.num_hidden_layers and not last_prenorm)
                                                      else int(nth), hip.ptr(out), hip.current_stream_ptr()))
        return out

    __call__ = encode
