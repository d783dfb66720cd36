"""`SAMAudioJudgeModel` / `PEAudioFrame` - host classes of the reranking and span-prediction rows
(reference sam_audio/model/judge.py:35-132, model.py:96-102,231-245).

What runs where
  * DAC-VAE encoder of the mixtures / separations: the codec kernels of the separate() path, through a codec-only
    engine context (`samaudio_finalize(ctx, 2)`).
  * both PE-AV transformers, the Judge's projections / LayerNorm / pooled head, the PE-A-Frame heads and frame logits:
    hand-written HIP behind `samaudio_judge_*` / `samaudio_frame_*` (include/samaudio.h).
  * the ModernBERT text tower: on the HIP library too (sam_audio_amd/mbert_encoder.py, `samaudio_mbert_*`); the `transformers`
    module only carries the weights.  The tokenizer is Hugging Face's (a once-per-call cost on a handful of tokens,
    SURVEY.md section 8 f1).
The reference repeats the mixture once per reranking candidate (ranking/judge.py:31-33).  Every op of the Judge is
per-row, so `score_candidates` evaluates the mixture branch once per clip: identical results, about half the DAC
encodes and transformer rows (pinned on the CPU by tests/test_judge_oracle.py::test_judge_rows_are_independent...).

State-dict keys: reference names for everything judge.py owns; the un-vendored PE-AV transformer uses the key names
of its Hugging Face port (transformers/models/pe_audio/modeling_pe_audio.py) - a documented assumption.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass
from typing import Any, Dict, List, Optional

import torch

from . import hip
from .config import PEAudioFrameConfig, PEAVTransformerConfig, SAMAudioJudgeConfig
from .judge_util import ensure_ws as _ensure_ws, register as _register
from .weights import _interleave16, convert_codec


@dataclass
class SAMAudioJudgeOutput:  # reference judge.py:16-32
    overall: Optional[torch.Tensor] = None
    recall: Optional[torch.Tensor] = None
    precision: Optional[torch.Tensor] = None
    faithfulness: Optional[torch.Tensor] = None
    text_model_output: Any = None
    audio_model_output: Any = None


# --------------------------------------------------------------------------------------------------------------
# weights
# --------------------------------------------------------------------------------------------------------------
def peav_keys(tc: PEAVTransformerConfig, prefix: str) -> List[str]:
    from .synthetic import init_peav_state_dict
    return list(init_peav_state_dict(tc, prefix, None, torch.device("meta")).keys())


def convert_peav(sd: Dict[str, torch.Tensor], src: str, dst: str, tc: PEAVTransformerConfig, in_w: torch.Tensor,
                 in_b: torch.Tensor, act_dtype: torch.dtype, device) -> Dict[str, torch.Tensor]:
    """One PE-AV transformer -> engine tensors under `dst` ("t." / "ft." / "a.").  Re-layouts: q|k|v fused (the HF
    port splits heads head-major, hf:381-385, so no row permutation is needed here - unlike the DiT's quirk Q1),
    gate|up interleaved in 16-row blocks for the SwiGLU epilogue, conv k3 weights tap-major, fp32 RoPE tables."""
    out: Dict[str, torch.Tensor] = {}

    def f32(x):
        return x.detach().to(device=device, dtype=torch.float32).contiguous()

    def op(x):
        return x.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()

    def W(key):
        return sd[src + key].detach().to(device=device, dtype=torch.float32)

    D = tc.hidden_size
    out[dst + "in.w"], out[dst + "in.b"] = op(in_w), f32(in_b)
    out[dst + "cls"] = f32(sd[src + "patch_embedder.class_embedding"].reshape(D))
    for n, blk in ((1, "block1"), (2, "block2")):
        B = f"patch_embedder.resnet_block.{blk}."
        out[f"{dst}gn{n}.w"], out[f"{dst}gn{n}.b"] = f32(sd[src + B + "groupnorm.weight"]), f32(sd[src + B + "groupnorm.bias"])
        out[f"{dst}conv{n}.w"] = op(W(B + "project.weight").permute(0, 2, 1).reshape(D, 3 * D))
        out[f"{dst}conv{n}.b"] = f32(sd[src + B + "project.bias"])
    for i in range(tc.num_hidden_layers):
        L, E = f"layers.{i}.", f"{dst}L{i}."
        out[E + "attn_norm"] = f32(sd[src + L + "input_layernorm.weight"])
        out[E + "ffn_norm"] = f32(sd[src + L + "post_attention_layernorm.weight"])
        out[E + "q_norm"] = f32(sd[src + L + "self_attn.q_norm.weight"])
        out[E + "k_norm"] = f32(sd[src + L + "self_attn.k_norm.weight"])
        out[E + "wqkv"] = op(torch.cat([W(L + f"self_attn.{n}_proj.weight") for n in "qkv"]))
        out[E + "wo"] = op(W(L + "self_attn.o_proj.weight"))
        if tc.attention_bias:
            out[E + "bqkv"] = f32(torch.cat([sd[src + L + f"self_attn.{n}_proj.bias"] for n in "qkv"]))
            out[E + "bo"] = f32(sd[src + L + "self_attn.o_proj.bias"])
        out[E + "w13"] = op(_interleave16(W(L + "mlp.gate_proj.weight"), W(L + "mlp.up_proj.weight")))
        out[E + "w2"] = op(W(L + "mlp.down_proj.weight"))
    out[dst + "norm"] = f32(sd[src + "norm.weight"])
    out[dst + "out.w"] = op(W("output.weight"))
    hd = tc.head_dim  # RoPE tables with the op sequence of hf:573-580,589-600 (fp32, adjacent pairs use freqs[:hd/2])
    inv_freq = 1.0 / (tc.rope_theta ** (torch.arange(0, hd, 2, dtype=torch.float) / hd))
    ang = torch.outer(torch.arange(tc.max_position_embeddings).float(), inv_freq)
    out[dst + "rope_cos"], out[dst + "rope_sin"] = f32(ang.cos()), f32(ang.sin())
    return out


def peav_dims(tc: PEAVTransformerConfig, in_dim: int) -> hip.PeavDims:
    return hip.PeavDims(dim=tc.hidden_size, n_heads=tc.num_attention_heads, n_layers=tc.num_hidden_layers,
                        ffn_hidden=tc.intermediate_size, in_dim=in_dim, max_positions=tc.max_position_embeddings,
                        attn_bias=int(tc.attention_bias), norm_eps=tc.rms_norm_eps)


JUDGE_OWN_KEYS = ("data_proj.weight", "data_proj.bias", "cat_audio_proj.weight", "cat_audio_proj.bias",
                  "text_proj1.weight", "text_proj2.weight", "text_proj2.bias", "layer_norm.weight", "layer_norm.bias",
                  "proj_audio_and_text.weight", "proj_audio_and_text.bias", "finetune_data_proj.weight",
                  "finetune_data_proj.bias", "head.weight", "mean", "std")


def convert_judge(sd: Dict[str, torch.Tensor], cfg: SAMAudioJudgeConfig, act_dtype: torch.dtype,
                  device) -> Dict[str, torch.Tensor]:
    def f32(x):
        return x.detach().to(device=device, dtype=torch.float32).contiguous()

    def op(x):
        return x.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()

    D, Bn = cfg.transformer.hidden_size, cfg.bottleneck_dim
    out = convert_peav(sd, "transformer.", "t.", cfg.transformer, sd["data_proj.weight"], sd["data_proj.bias"],
                       act_dtype, device)
    out.update(convert_peav(sd, "finetune_transformer.", "ft.", cfg.finetune_transformer,
                            sd["finetune_data_proj.weight"], sd["finetune_data_proj.bias"], act_dtype, device))
    cat = sd["cat_audio_proj.weight"]          # input = cat([hyp_features, input_features]) (judge.py:113-115)
    out["cat.wh"], out["cat.wi"], out["cat.b"] = op(cat[:, :D]), op(cat[:, D:]), f32(sd["cat_audio_proj.bias"])
    out["tp1.w"] = op(sd["text_proj1.weight"])
    out["tp2.w"], out["tp2.b"] = op(sd["text_proj2.weight"]), f32(sd["text_proj2.bias"])
    out["ln.w"], out["ln.b"] = f32(sd["layer_norm.weight"]), f32(sd["layer_norm.bias"])
    pat = sd["proj_audio_and_text.weight"]     # input = cat([audio_features, expanded_text]) (judge.py:121-123)
    out["pat.wa"], out["pat.wt"], out["pat.b"] = op(pat[:, :Bn]), op(pat[:, Bn:]), f32(sd["proj_audio_and_text.bias"])
    out["head.w"] = f32(sd["head.weight"])
    out["mean"], out["std"] = f32(sd["mean"].reshape(4)), f32(sd["std"].reshape(4))
    return out


class _CodecEncoder:
    """Codec-only engine context: the DACVAEEncoder of reference codec.py:42-78 on the separate() path's kernels."""

    def __init__(self, codec_cfg, precision: str, device: torch.device):
        self.cfg, self.device = codec_cfg, device
        self._lib = hip.lib(hip.operands_for(precision))
        self._ctx = C.c_void_p()
        hc = hip.Config(
            precision=hip.precision_code(precision), dim=256, n_heads=2, n_layers=0, ffn_hidden=64,
            latent_channels=2 * codec_cfg.codebook_dim, text_dim=64, video_dim=64, freq_dim=64, anchor_dim=64,
            anchor_vocab=4, max_positions=64, norm_eps=1e-5, codec_dim=codec_cfg.codebook_dim,
            codec_latent=codec_cfg.latent_dim, enc_dim=codec_cfg.encoder_dim, dec_dim=codec_cfg.decoder_dim,
            enc_rates=(C.c_int32 * 4)(*codec_cfg.encoder_rates), dec_rates=(C.c_int32 * 4)(*codec_cfg.decoder_rates))
        hip.check(self._lib.samaudio_create(C.byref(hc), C.byref(self._ctx)))
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None

    def __del__(self):
        if getattr(self, "_ctx", None):
            self._lib.samaudio_destroy(self._ctx)
            self._ctx = None

    def load(self, tensors: Dict[str, torch.Tensor]) -> None:
        for name, t in tensors.items():
            dt = hip.dtype_code(t.dtype)
            if t.data_ptr() % 16:
                t = t.clone()
            self._tensors[name] = t
            hip.check(self._lib.samaudio_set_tensor(self._ctx, name.encode(), hip.ptr(t), dt, t.dim(),
                                                    hip.shape_array(t.shape)))
        hip.check(self._lib.samaudio_finalize(self._ctx, 2))

    def encode(self, audios: torch.Tensor) -> torch.Tensor:
        """[N, 1, Tw] -> mean latents, channels-last [N, T, codebook_dim] (right reflect-pad to the hop, codec.py:72-78)."""
        hop = self.cfg.hop_length
        wav = audios.to(self.device, torch.float32)
        rem = wav.size(-1) % hop
        if rem:
            wav = torch.nn.functional.pad(wav, (0, hop - rem), mode="reflect")
        wav = wav.squeeze(1).contiguous()
        items, samples = wav.shape
        z = torch.empty(items, samples // hop, self.cfg.codebook_dim, device=self.device)
        chunk = min(items, int(os.environ.get("SAMAUDIO_CODEC_CHUNK", "16")))
        need = self._lib.samaudio_workspace_bytes(self._ctx, 0, 0, 1, chunk, samples)
        _ensure_ws(self, need, lambda p, n: self._lib.samaudio_set_workspace(self._ctx, p, n))
        hip.check(self._lib.samaudio_codec_encode(self._ctx, hip.ptr(wav), items, samples, hip.ptr(z),
                                                  hip.current_stream_ptr()))
        return z


class _TextTower:
    """The ModernBERT text tower of a Judge / PE-A-Frame model.  `module` (a `transformers.ModernBertModel`) is the CONTAINER
    of the weights - `state_dict()` / `load_state_dict()` keep their reference semantics - and is never executed: the forward
    runs on the HIP library (sam_audio_amd/mbert_encoder.py; fp32: a handful of tokens once per call) once the model sits on
    a GPU.  (The tests' checker runs the module itself: tests/torch_text.py.)"""

    def __init__(self, module):
        self.module = module
        self._hip = None
        self._device = None

    def place(self, device) -> None:
        """(re)build the device copy after the weights changed or the model moved"""
        self._device = torch.device(device)
        from .mbert_encoder import ModernBertHIP
        self._hip = ModernBertHIP.from_module(self.module, self._device, precision="fp32")

    def hidden(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], nth: Optional[int],
               last_prenorm: bool = True) -> torch.Tensor:
        """transformers' `hidden_states[nth]` ([B, Lt, hidden]); nth None = last_hidden_state.  nth == num_hidden_layers is
        version-dependent in transformers (SAMAudioJudgeConfig.last_text_layer_prenorm): `last_prenorm` True = the 4.x
        meaning (output of the last layer, before final_norm), False = the 5.x meaning (last_hidden_state) - independent of
        the transformers version that happens to be installed."""
        if self._hip is None:
            raise hip.SamAudioHipError("text tower: load the model's weights on a ROCm GPU first (no CPU fallback)")
        return self._hip(input_ids, attention_mask, nth, last_prenorm=last_prenorm)


def _text_tower(text_cfg: Dict[str, Any]):
    import transformers
    return transformers.AutoModel.from_config(transformers.ModernBertConfig(**text_cfg)).eval()


# --------------------------------------------------------------------------------------------------------------
# Judge
# --------------------------------------------------------------------------------------------------------------
class SAMAudioJudgeModel:
    config_cls = SAMAudioJudgeConfig

    def __init__(self, config: SAMAudioJudgeConfig, precision: str = "bf16", device: Optional[str] = None,
                 text_model=None):
        config.check_supported()
        hip.check_precision(precision)
        self.config = config
        self.precision = precision
        self.device = torch.device(device) if device is not None else None
        self.text_model = text_model if text_model is not None else _text_tower(config.text_model)  # judge.py:48
        self._text = _TextTower(self.text_model)
        self._lib = hip.lib(hip.operands_for(precision))
        self._h = C.c_void_p()
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self._codec: Optional[_CodecEncoder] = None
        self._loaded = False
        jc = hip.JudgeConfig(
            precision=hip.precision_code(precision),
            transformer=peav_dims(config.transformer, config.audio_codec.codebook_dim),
            finetune_transformer=peav_dims(config.finetune_transformer, config.bottleneck_dim),
            codec_dim=config.audio_codec.codebook_dim, text_hidden=config.text_hidden,
            bottleneck_dim=config.bottleneck_dim)
        hip.check(self._lib.samaudio_judge_create(C.byref(jc), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.samaudio_judge_destroy(self._h)
            self._h = None

    @property
    def act_dtype(self) -> torch.dtype:
        return hip.act_dtype(self.precision)

    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if self._tensors and self.device != device:
            raise RuntimeError("move the model before load_state_dict (weights are converted onto the device)")
        self.device = device
        return self

    def cuda(self, index: int = 0):
        return self.to(f"cuda:{index}")

    @classmethod
    def from_pretrained(cls, model_id: str, map_location: str = "cpu", strict: bool = True, precision: str = "bf16",
                        device: Optional[str] = None, **model_kwargs):
        """Local directory with `config.json` + `checkpoint.pt` (reference base.py:17-62)."""
        if not os.path.isdir(model_id):
            raise FileNotFoundError(f"{model_id}: only local checkpoint directories are supported offline")
        with open(os.path.join(model_id, "config.json")) as fin:
            config = json.load(fin)
        for key, value in model_kwargs.items():
            if key in config:
                config[key] = value
        model = cls(SAMAudioJudgeConfig(**config), precision=precision, device=device)
        sd = torch.load(os.path.join(model_id, "checkpoint.pt"), weights_only=True, map_location=map_location)
        model.load_state_dict(sd, strict=strict)
        return model

    def expected_keys(self, with_codec: bool = True, with_text: bool = True) -> List[str]:
        cfg = self.config
        keys = list(JUDGE_OWN_KEYS) + peav_keys(cfg.transformer, "transformer.") + \
            peav_keys(cfg.finetune_transformer, "finetune_transformer.")
        if with_codec:
            from .synthetic import _encoder_only_codec
            keys += list(_encoder_only_codec(cfg.audio_codec, None, torch.device("meta")).keys())
        if with_text:
            keys += ["text_model." + k for k in self.text_model.state_dict().keys()]
        return keys

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Reference key names in (torch.nn.Module.load_state_dict semantics: strict raises RuntimeError)."""
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hip.require_gpu(self.device, "SAMAudioJudgeModel")
        import re
        canon = lambda k: re.sub(r"\.(weight_g|weight_v|parametrizations\.weight\.original[01])$", ".weight", k)  # noqa: E731
        have = {canon(k) for k in state_dict}
        want = set(self.expected_keys())
        # the reference DACVAEEncoder keeps the WHOLE quantizer of the dacvae model (codec.py:62-63: in_proj, out_proj, ...),
        # and a checkpoint saved from a full DACVAE also carries the decoder: present in a genuine checkpoint, unused here
        unused = re.compile(r"^audio_codec\.(quantizer\.(?!in_proj\.)|decoder\.)")
        missing = sorted(want - have)
        unexpected = sorted(k for k in state_dict if canon(k) not in want and not unused.search(k))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Missing keys: {missing}, unexpected_keys: {unexpected}")
        text_sd = {k[len("text_model."):]: v for k, v in state_dict.items() if k.startswith("text_model.")}
        if text_sd:
            self.text_model.load_state_dict(text_sd, strict=False)
        self._text.place(self.device)
        self.text_model = self._text.module
        with torch.cuda.device(self.device):
            if not any(k for k in missing if not k.startswith(("audio_codec.", "text_model."))):
                _register(self._lib.samaudio_judge_set_tensor, self._h, self._tensors,
                          convert_judge(state_dict, self.config, self.act_dtype, self.device))
                hip.check(self._lib.samaudio_judge_finalize(self._h))
                self._loaded = True
            if not any(k for k in missing if k.startswith("audio_codec.")):
                self._codec = _CodecEncoder(self.config.audio_codec, self.precision, self.device)
                self._codec.load(convert_codec(state_dict, self.config, self.act_dtype, self.device, with_decoder=False))
        return missing, unexpected

    # ---------------------------------------------------------------------------------------------- forward
    @torch.inference_mode()
    def _get_text_output(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
        """reference judge.py:76-88 -> the n-th hidden state [B, Lt, hidden] (pooler_output = [:, 0])."""
        return self._text.hidden(input_ids, attention_mask, self.config.nth_text_layer,
                                 last_prenorm=self.config.last_text_layer_prenorm)

    def _score(self, in_lat: torch.Tensor, sep_lat: torch.Tensor, cand: int, pooled: torch.Tensor,
               frame_mask: Optional[torch.Tensor]) -> torch.Tensor:
        inputs, frames, _ = in_lat.shape
        assert sep_lat.shape[0] == inputs * cand and sep_lat.shape[1] == frames, "separated/mixture shape mismatch"
        assert pooled.shape == (inputs * cand, self.config.text_hidden), "one text row per (clip, candidate) pair"
        pooled = pooled.to(self.device, torch.float32).contiguous()
        mask = None if frame_mask is None else frame_mask.to(self.device).to(torch.uint8).contiguous()
        scores = torch.empty(inputs * cand, 4, device=self.device)
        need = self._lib.samaudio_judge_workspace_bytes(self._h, inputs, cand, frames)
        _ensure_ws(self, need, lambda p, n: self._lib.samaudio_judge_set_workspace(self._h, p, n))
        hip.check(self._lib.samaudio_judge_score(self._h, hip.ptr(in_lat), hip.ptr(sep_lat), inputs, cand, frames,
                                                 hip.ptr(pooled), hip.ptr(mask), hip.ptr(scores),
                                                 hip.current_stream_ptr()))
        return scores

    def _frame_mask(self, padding_mask: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        if padding_mask is None:
            return None
        return padding_mask[:, :: self.config.audio_codec.hop_length]  # judge.py:104-107

    @torch.inference_mode()
    def forward(self, input_ids: torch.Tensor, input_values: torch.Tensor, separated_values: torch.Tensor,
                attention_mask: Optional[torch.Tensor] = None,
                padding_mask: Optional[torch.Tensor] = None) -> SAMAudioJudgeOutput:
        """reference judge.py:90-132 (same arguments): input_values / separated_values [B, 1, Tw], padding_mask
        [B, Tw] bool."""
        if not (self._loaded and self._codec is not None):
            raise RuntimeError("load_state_dict() first")
        with torch.cuda.device(self.device):
            hidden = self._get_text_output(input_ids, attention_mask)
            lat = self._codec.encode(torch.cat([input_values, separated_values], dim=0))        # judge.py:101-102
            B = input_values.shape[0]
            scores = self._score(lat[:B].contiguous(), lat[B:].contiguous(), 1, hidden[:, 0].float(),
                                 self._frame_mask(padding_mask))
        return SAMAudioJudgeOutput(*scores.chunk(4, dim=1), text_model_output=hidden)

    __call__ = forward

    @torch.inference_mode()
    def score_candidates(self, input_ids: torch.Tensor, input_values: torch.Tensor, separated_values: torch.Tensor,
                         candidates: int, attention_mask: Optional[torch.Tensor] = None,
                         padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reranker's batch shape: `input_values` [B, 1, Tw] (one mixture per clip), `separated_values`
        [B*candidates, 1, Tw] sample-major, `input_ids` [B, Lt] (one description per clip), padding_mask [B, Tw].
        Returns the `overall` scores [B, candidates] (ranking/judge.py:41-42)."""
        if not (self._loaded and self._codec is not None):
            raise RuntimeError("load_state_dict() first")
        with torch.cuda.device(self.device):
            pooled = self._get_text_output(input_ids, attention_mask)[:, 0].float()
            pooled = pooled.repeat_interleave(candidates, dim=0)
            B = input_values.shape[0]
            lat = self._codec.encode(torch.cat([input_values.to(self.device), separated_values.to(self.device)], dim=0))
            scores = self._score(lat[:B].contiguous(), lat[B:].contiguous(), candidates, pooled,
                                 self._frame_mask(padding_mask))
        return scores[:, 0].view(B, candidates)


# --------------------------------------------------------------------------------------------------------------
# PE-A-Frame span predictor
# --------------------------------------------------------------------------------------------------------------
SPAN_GUARD_S = 1e-6


def spans_from_logits(logits: torch.Tensor, pad_mask: Optional[torch.Tensor], hop: int, sample_rate: int,
                      threshold: float = 0.5) -> List[List[List[float]]]:
    """Frame logits -> [[start_s, end_s], ...] per row.  The reference's rule (`return_spans=True` of the
    un-vendored PEAudioFrame) is not reachable offline; this build DEFINES it as: a frame is active iff
    sigmoid(logit) > threshold and it is a valid frame; every maximal run [s, e) of active frames is one span
    [s*hop/sr - 1 us, e*hop/sr - 1 us] (start clamped at 0).  The microsecond guard makes `Batch.process_anchors`
    (reference processor.py:107-121, ceil(seconds*sr/hop)) map the span back onto exactly frames [s, e).
    Integer work on a few hundred frames per clip: host side, like process_anchors itself."""
    import math
    cut = math.log(threshold / (1.0 - threshold))
    active = logits > cut
    if pad_mask is not None:
        active = active & pad_mask.to(active.device)
    out: List[List[List[float]]] = []
    for row in active.cpu().tolist():
        spans, start = [], None
        for t, on in enumerate(row + [False]):
            if on and start is None:
                start = t
            elif not on and start is not None:
                spans.append([max(0.0, start * hop / sample_rate - SPAN_GUARD_S), t * hop / sample_rate - SPAN_GUARD_S])
                start = None
        out.append(spans)
    return out


def convert_frame(sd: Dict[str, torch.Tensor], cfg: PEAudioFrameConfig, act_dtype: torch.dtype,
                  device) -> Dict[str, torch.Tensor]:
    def f32(x):
        return x.detach().to(device=device, dtype=torch.float32).contiguous()

    def op(x):
        return x.detach().to(device=device, dtype=torch.float32).to(act_dtype).contiguous()

    out = convert_peav(sd, "audio_encoder.", "a.", cfg.audio, sd["audio_encoder.embedder.data_proj.weight"],
                       sd["audio_encoder.embedder.data_proj.bias"], act_dtype, device)
    out["ah.ln_w"], out["ah.ln_b"] = f32(sd["audio_head.layer_norm.weight"]), f32(sd["audio_head.layer_norm.bias"])
    out["ah.w"] = op(sd["audio_head.proj.weight"])
    out["th.ln_w"], out["th.ln_b"] = f32(sd["text_audio_head.layer_norm.weight"]), f32(sd["text_audio_head.layer_norm.bias"])
    out["th.w"] = op(sd["text_audio_head.proj.weight"])
    out["logit_scale"] = f32(sd["text_audio_logit_scale"].reshape(1))
    out["logit_bias"] = f32(sd["text_audio_logit_bias"].reshape(1))
    return out


@dataclass
class PEAudioFrameOutput:
    logits: torch.Tensor                      # [B, T]
    spans: Optional[List[List[List[float]]]]  # per row: [[start_s, end_s], ...]


class PEAudioFrame:
    """Span predictor with the call shape of reference model.py:234-243:
    `predictor(input_features=[B, T, 128], padding_mask=[B, T], return_spans=True, input_ids=..., attention_mask=...)`."""

    def __init__(self, config: PEAudioFrameConfig, precision: str = "bf16", device: Optional[str] = None,
                 text_model=None, hop_length: int = 1920, sample_rate: int = 48_000):
        config.check_supported()
        self.config, self.precision = config, precision
        self.device = torch.device(device) if device is not None else None
        self.hop_length, self.sample_rate = hop_length, sample_rate
        self.text_model = text_model if text_model is not None else _text_tower(config.text_model)
        self._text = _TextTower(self.text_model)
        self._lib = hip.lib(hip.operands_for(precision))
        self._h = C.c_void_p()
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self._loaded = False
        fc = hip.FrameConfig(precision=hip.precision_code(precision),
                             audio=peav_dims(config.audio, config.codebook_dim), codec_dim=config.codebook_dim,
                             embed_dim=config.text_hidden)
        hip.check(self._lib.samaudio_frame_create(C.byref(fc), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.samaudio_frame_destroy(self._h)
            self._h = None

    @property
    def act_dtype(self) -> torch.dtype:
        return hip.act_dtype(self.precision)

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hip.require_gpu(self.device, "PEAudioFrame")
        text_sd = {k[len("text_model."):]: v for k, v in state_dict.items() if k.startswith("text_model.")}
        if text_sd:
            self.text_model.load_state_dict(text_sd, strict=strict)
        self._text.place(self.device)
        self.text_model = self._text.module
        with torch.cuda.device(self.device):
            _register(self._lib.samaudio_frame_set_tensor, self._h, self._tensors,
                      convert_frame(state_dict, self.config, self.act_dtype, self.device))
            hip.check(self._lib.samaudio_frame_finalize(self._h))
        self._loaded = True

    @torch.inference_mode()
    def frame_logits(self, input_features: torch.Tensor, text_pooled: torch.Tensor,
                     padding_mask: Optional[torch.Tensor]) -> torch.Tensor:
        if not self._loaded:
            raise RuntimeError("load_state_dict() first")
        feats = input_features.to(self.device, torch.float32).contiguous()
        rows, frames, width = feats.shape
        assert width == self.config.codebook_dim, "input_features must be the codec mean latent [B, T, codebook_dim]"
        pooled = text_pooled.to(self.device, torch.float32).contiguous()
        assert pooled.shape == (rows, self.config.text_hidden)
        mask = None if padding_mask is None else padding_mask.to(self.device).to(torch.uint8).contiguous()
        logits = torch.empty(rows, frames, device=self.device)
        with torch.cuda.device(self.device):
            need = self._lib.samaudio_frame_workspace_bytes(self._h, rows, frames)
            _ensure_ws(self, need, lambda p, n: self._lib.samaudio_frame_set_workspace(self._h, p, n))
            hip.check(self._lib.samaudio_frame_logits(self._h, hip.ptr(feats), hip.ptr(pooled), hip.ptr(mask), rows,
                                                      frames, hip.ptr(logits), hip.current_stream_ptr()))
        return logits

    @torch.inference_mode()
    def __call__(self, input_features: torch.Tensor, padding_mask: Optional[torch.Tensor] = None,
                 return_spans: bool = False, input_ids: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None,
                 text_pooled: Optional[torch.Tensor] = None) -> PEAudioFrameOutput:
        if text_pooled is None:
            text_pooled = self._text.hidden(input_ids, attention_mask, None)[:, 0].float()  # hf:847: hidden_states[-1][:, 0]
        logits = self.frame_logits(input_features, text_pooled, padding_mask)
        spans = None
        if return_spans:
            spans = spans_from_logits(logits, padding_mask, self.hop_length, self.sample_rate, self.config.threshold)
        return PEAudioFrameOutput(logits=logits, spans=spans)
