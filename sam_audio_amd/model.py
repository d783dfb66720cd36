"""`SAMAudio` - drop-in host class for the reference's SAMAudio.separate() path
(reference sam_audio/model/model.py:75-359), backed by libsamaudio_hip.so.

PyTorch-ROCm is used for device memory, streams, RNG and weight plumbing only; every arithmetic step
of separate() - DAC-VAE encode, the 32 DiT evaluations of the midpoint ODE, DAC-VAE decode - runs in
hand-written HIP kernels behind the C ABI of include/samaudio.h.  No eager/CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional

import torch

from . import hip
from .config import SAMAudioConfig
from .processor import Batch
from .weights import (F32_SOURCE_KEYS, convert_codec, convert_codec_fly16, convert_codec_x3, convert_dit, convert_dit_f32, convert_dit_x3,
                      split_missing_unexpected)

DFLT_ODE_OPT = {"method": "midpoint", "options": {"step_size": 2 / 32}}  # reference model.py:22
# Layout of the five big DiT weight matrices and the next-weights prefetch of few-row launches (SAMAudio.__init__; DESIGN.md
# section 7 has the measurements behind the defaults).  Environment overrides (tuning): SAMAUDIO_WEIGHT_LAYOUT,
# SAMAUDIO_PREFETCH_ROWS.
DEFAULT_WEIGHT_LAYOUT = "ktm"
DEFAULT_PREFETCH_ROWS = 2048


@dataclass
class SeparationResult:  # reference model.py:68-72
    target: List[torch.Tensor]
    residual: List[torch.Tensor]
    noise: torch.Tensor


def ode_grid(ode_opt: Dict[str, Any], t0: float = 0.0, t1: float = 1.0):
    """(method id, grid) for the fixed-grid solvers torchdiffeq would run for `ode_opt`
    (reference model.py:285-290; quirk Q11).  Anything but midpoint/euler + step_size is rejected."""
    method = ode_opt.get("method", "midpoint")
    if method not in ("midpoint", "euler"):
        raise ValueError(f"ode method {method!r} is not supported by the HIP path (midpoint | euler)")
    options = dict(ode_opt.get("options", {}))
    step = options.pop("step_size", None)
    if step is None or options:
        raise ValueError("ode_opt['options'] must be exactly {'step_size': float}")
    n = int(math.ceil((t1 - t0) / step + 1))
    grid = [min(t0 + k * step, t1) for k in range(n)]
    grid[-1] = t1
    return (hip.ODE_MIDPOINT if method == "midpoint" else hip.ODE_EULER), grid


class _Codec16:
    """A codec-only engine context on plain 16-bit operands beside an fp32-storage model (precision "fp16x3", `codec_decode="16"`):
    SAMAudio.decode_audio runs on it.  It borrows the model's 16-bit copies of the codec weights and whatever workspace its owner
    (the model or a stream lane) holds at the time of the call."""

    def __init__(self, model: "SAMAudio"):
        self.lib = model._lib
        self._ctx = C.c_void_p()
        hc = hip.Config.from_buffer_copy(model._hc)
        hc.precision = hip.BF16   # "the library's 16-bit operand format"
        hip.check(self.lib.samaudio_create(C.byref(hc), C.byref(self._ctx)))
        for name, t in model._codec16_tensors.items():
            dt = hip.dtype_code(t.dtype, hip.operands_for(model.precision))
            hip.check(self.lib.samaudio_set_tensor(self._ctx, name.encode(), hip.ptr(t), dt, t.dim(), hip.shape_array(t.shape)))
        hip.check(self.lib.samaudio_finalize(self._ctx, 1))

    def __del__(self):
        if getattr(self, "_ctx", None):
            self.lib.samaudio_destroy(self._ctx)
            self._ctx = None


class _Lane:
    """An extra engine context bound to its own HIP stream.  It borrows the model's weight tensors (no copy) and
    owns only a workspace; `SAMAudio(streams=S)` solves S contiguous row groups of a batch concurrently so that one
    group's GEMM tails / epilogues / launch gaps are filled by the other group's workgroups."""

    def __init__(self, model: "SAMAudio"):
        self.lib = model._lib
        self._ctx = C.c_void_p()
        hip.check(self.lib.samaudio_create(C.byref(model._hc), C.byref(self._ctx)))
        for name, t in model._tensors.items():
            dt = hip.dtype_code(t.dtype, hip.operands_for(model.precision), alt_ok=model._alt16_weight(name))
            hip.check(self.lib.samaudio_set_tensor(self._ctx, name.encode(), hip.ptr(t), dt, t.dim(),
                                                   hip.shape_array(t.shape)))
        hip.check(self.lib.samaudio_finalize(self._ctx, 0))
        model._set_precision_options(self._ctx)
        if model._has_codec:   # the lane also decodes its own row group (SAMAudio._solve_concurrent)
            hip.check(self.lib.samaudio_finalize(self._ctx, 1))
        self._workspace: Optional[torch.Tensor] = None
        self._live = None
        self._codec16: Optional[_Codec16] = None
        self.stream = torch.cuda.Stream(device=model.device)

    def __del__(self):
        if getattr(self, "_ctx", None):
            self.lib.samaudio_destroy(self._ctx)
            self._ctx = None


class SAMAudio:
    config_cls = SAMAudioConfig

    def __init__(self, cfg: SAMAudioConfig, precision: str = "fp16x3", device: Optional[str] = None,
                 text_encoder: Optional[Callable] = None, streams: int = 1, f32_classes="auto",
                 weight_layout: str = "auto", prefetch_rows: Optional[int] = None, x3_classes="auto",
                 codec_decode: str = "auto"):
        """`precision`: "fp16x3" (default: the reference computes in fp32, README.md:48 - fp32 storage and every big contraction on
        hi/lo-split IEEE-half operands hold its results to 1e-3 on benign and trained-like weights) | "fp16" |
        "mixed" | "bf16" (plain 16-bit GEMM operands: ~3x the throughput, inside 1e-3 on benign weights only / not at all) | "fp32"
        (exact-fp32 MFMA) | "bf16x3".
        `weight_layout` (16-bit precisions): "ktm" stores the weights of the five big GEMM classes of the DiT layers
        K-tile-major (weights.ktm_layout; a launch then streams its weights front to back), "rows" keeps them row-major;
        "auto" = DEFAULT_WEIGHT_LAYOUT.  `prefetch_rows`: evaluations of at most this many rows (batch x frames) let the CUs a
        GEMM launch leaves idle read the next GEMM's weights (samaudio.h SAMAUDIO_OPT_PREFETCH_ROWS; None =
        DEFAULT_PREFETCH_ROWS, 0 = off).  Both are scheduling / layout choices: results are bitwise the same.
        `f32_classes` (16-bit precisions): the GEMM classes that run on exact-fp32 operands inside the 16-bit engine
        (hip.CLASSES names or a mask; "auto" = hip.CLS_F32_DEFAULT: the input and output projections that touch the ODE
        state and the hoisted conditioning, which carry most of a 16-bit mode's error for < 1 % of a step; "time" and
        "yemb" can be added - DESIGN.md section 4)."""
        cfg.check_supported()
        hip.check_precision(precision, x3_ok=True)
        self.cfg = cfg
        self.precision = precision
        # precision "fp16x3" (and "bf16x3"): fp32 storage and fp32 small classes, the six big GEMM classes of the layers on
        # compensated 16-bit operands (samaudio.h SAMAUDIO_OPT_X3_CLASSES; `x3_classes`: names or a mask, "auto" = all six)
        self.x3_classes = 0 if not hip.is_x3(precision) else (
            hip.CLS_X3_DEFAULT if x3_classes == "auto" else hip.class_mask(x3_classes))
        # x3 precisions: the DAC-VAE DECODER in the model's own fp32 context ("32": its convolutions multiply on operands split on
        # the fly like the encoder's - hip.CLS["codec"] of x3_classes) or on plain 16-bit operands ("16": a codec-only context of
        # the same library beside the fp32 one).  Measured on the hostile weights (profiles/r6_call2/): with
        # the 16-bit decoder the waveform misses the 1e-3 bound (1.6e-3 small*, 1.9e-3 large*; latent 1e-4 either way), so "32" is
        # the default; "16" stays selectable (benign weights: 2e-4).  "auto" = environment SAMAUDIO_X3_DECODE, default "32".
        if codec_decode == "auto":
            codec_decode = os.environ.get("SAMAUDIO_X3_DECODE", "32")
        if codec_decode not in ("16", "32"):
            raise ValueError("codec_decode must be 'auto', '16' or '32'")
        self.codec_decode = codec_decode if hip.is_x3(precision) else "native"
        self._codec16_tensors: Dict[str, torch.Tensor] = {}
        self._codec16: Optional[_Codec16] = None
        precision = hip.storage_precision(precision)   # what everything below means by "fp32"
        self.f32_classes = 0 if precision == "fp32" else (
            hip.CLS_F32_DEFAULT if f32_classes == "auto" else hip.class_mask(f32_classes))
        self.quant_classes, self.quant_format = 0, 0   # fp32 engines: operand-rounding emulation (error budget)
        # precision="mixed": bf16 operands for the five big GEMM classes inside the fp16 build (hip.CLS_ALT16_MIXED)
        self.alt16_classes = hip.CLS_ALT16_MIXED if precision == "mixed" else 0
        if weight_layout == "auto":
            weight_layout = os.environ.get("SAMAUDIO_WEIGHT_LAYOUT", DEFAULT_WEIGHT_LAYOUT)
        if weight_layout not in ("ktm", "rows"):
            raise ValueError("weight_layout must be 'auto', 'ktm' or 'rows'")
        self.weight_layout = "rows" if precision == "fp32" else weight_layout
        if prefetch_rows is None:
            prefetch_rows = int(os.environ.get("SAMAUDIO_PREFETCH_ROWS", DEFAULT_PREFETCH_ROWS))
        self.prefetch_rows = 0 if precision == "fp32" else int(prefetch_rows)
        self.device = torch.device(device) if device is not None else None
        self.text_encoder = text_encoder      # callable: list[str] -> (features [B,Lt,768], mask [B,Lt])
        # rerankers (reference model.py:94-95): any callable with the reference's Ranker.forward keywords that returns
        # scores [B, candidates]; sam_audio_amd.ranking.JudgeRanker is the HIP-backed one.  Configured rankers are
        # built lazily from LOCAL checkpoint directories only (no hub access offline), see attach_rankers().
        self.visual_ranker = None
        self.text_ranker = None
        # span predictor (reference model.py:96-102): a sam_audio_amd.judge.PEAudioFrame plus the tokenizer-like
        # transform that turns descriptions into its text inputs
        self.span_predictor = None
        self.span_predictor_transform = None
        self.vision_encoder = None            # callable: list of [T,3,H,W] videos -> [B, T, vision_encoder.dim]
        self.fix_span_order = False           # quirk Q13, see separate()
        self._lib = hip.lib(hip.operands_for(self.precision))   # raises if the HIP library is not built
        self._ctx = C.c_void_p()
        self._tensors: Dict[str, torch.Tensor] = {}
        self._workspace: Optional[torch.Tensor] = None
        self._has_dit = self._has_codec = False
        if int(streams) not in (1, 2) and not os.environ.get("SAMAUDIO_ALLOW_STREAMS"):
            # measured on MI355X: 2 groups +3 %, 3-4 groups no further gain
            raise ValueError("streams must be 1 or 2 (set SAMAUDIO_ALLOW_STREAMS=1 to experiment with more)")
        self.streams = int(streams)           # row groups solved concurrently on separate HIP streams
        # GEMM launches split into whole rounds + a small-tile tail (samaudio.h SAMAUDIO_OPT_TAIL_SPLIT): None = automatic,
        # on when the batch is solved as one row group, off when two groups share the GPU (their kernels fill each other's
        # tails: +2 % without the split, profiles/r2_call7/); True / False pin it (bench.py keeps the instrumented step on
        # the same kernels as the timed ones)
        self.tail_split: Optional[bool] = None
        self._lanes: List[_Lane] = []
        self._f32_have = 0                    # F32-capable classes whose fp32 operand copies are registered
        self._f32_sd: Dict[str, torch.Tensor] = {}
        self._profiling = self._serial_groups = False
        self._sentinel = False
        t, c = cfg.transformer, cfg.audio_codec
        hc = hip.Config(
            precision=hip.precision_code(self.precision), dim=t.dim, n_heads=t.n_heads,
            n_layers=t.n_layers, ffn_hidden=t.ffn_hidden, latent_channels=t.out_channels,
            text_dim=cfg.text_encoder.dim, video_dim=cfg.vision_encoder.dim, freq_dim=t.frequency_embedding_dim,
            anchor_dim=cfg.anchor_embedding_dim, anchor_vocab=cfg.num_anchors + 1, max_positions=t.max_positions,
            norm_eps=t.norm_eps, codec_dim=c.codebook_dim, codec_latent=c.latent_dim, enc_dim=c.encoder_dim,
            dec_dim=c.decoder_dim, enc_rates=(C.c_int32 * 4)(*c.encoder_rates),
            dec_rates=(C.c_int32 * 4)(*c.decoder_rates))
        self._hc = hc
        hip.check(self._lib.samaudio_create(C.byref(hc), C.byref(self._ctx)))
        self._set_precision_options(self._ctx)

    def __del__(self):
        ctx = getattr(self, "_ctx", None)
        if ctx:
            self._lib.samaudio_destroy(ctx)
            self._ctx = None

    # ------------------------------------------------------------------ nn.Module-ish surface
    @property
    def act_dtype(self) -> torch.dtype:
        return hip.act_dtype(self.precision)

    @property
    def sample_rate(self) -> int:  # reference model.py:104-106
        return self.cfg.audio_codec.sample_rate

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
    def from_pretrained(cls, model_id: str, map_location: str = "cpu", strict: bool = True,
                        precision: str = "fp16x3", device: Optional[str] = None, **model_kwargs):
        """Local directory with the reference's `config.json` + `checkpoint.pt`
        (reference base.py:17-62; hub download needs network access this build does not have)."""
        if not os.path.isdir(model_id):
            raise FileNotFoundError(f"{model_id}: only local checkpoint directories are supported offline")
        with open(os.path.join(model_id, "config.json")) as fin:
            config = json.load(fin)
        for key, value in model_kwargs.items():
            if key in config:
                config[key] = value
        model = cls(SAMAudioConfig(**config), precision=precision, device=device)
        sd = torch.load(os.path.join(model_id, "checkpoint.pt"), weights_only=True, map_location=map_location)
        model.load_state_dict(sd, strict=strict)
        # the reference builds its T5 encoder and rankers in __init__ from hub ids (model.py:82,94-95); offline they
        # can only come from local directories - attach what is reachable, leave the rest to the caller
        try:   # a directory, or a hub id that is already in the local Hugging Face cache (local_files_only)
            from .text_encoder import T5TextEncoder
            model.text_encoder = T5TextEncoder(model.cfg.text_encoder, device=model.device)
        except FileNotFoundError as exc:
            warnings.warn(f"text encoder not attached ({exc}); pass text_features / text_mask to the processor or set "
                          "model.text_encoder")
        model.attach_rankers(precision=hip.tower_precision(precision))
        return model

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Reference key names in, engine tensors out (see weights.py).  Missing text-encoder / ranker /
        span-predictor keys are tolerated exactly like reference model.py:346-359."""
        if self.device is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        hip.require_gpu(self.device, "SAMAudio (the separate() hot path)")
        missing, unexpected = split_missing_unexpected(state_dict.keys(), self.cfg)
        codec_missing = [k for k in missing if k.startswith("audio_codec.")]
        dit_missing = [k for k in missing if not k.startswith("audio_codec.")]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Missing keys: {missing}, unexpected_keys: {unexpected}")
        vis = {k[len("vision_encoder."):]: v for k, v in state_dict.items() if k.startswith("vision_encoder.")}
        if vis and self.vision_encoder is None and self._covers_vision_tower(vis):
            # reference model.py:83: SAMAudio owns `PerceptionEncoder(cfg.vision_encoder)`; build it when the checkpoint
            # really carries the PE-Core tower (a text-only deployment pays nothing for it)
            from .vision_encoder import PerceptionEncoder
            self.vision_encoder = PerceptionEncoder(self.cfg.vision_encoder, device=self.device,
                                                    precision=hip.tower_precision(self.precision))
        if vis and hasattr(self.vision_encoder, "load_state_dict"):
            # The tower's key list is restated from the published PE-Core architecture (perception_models is not
            # importable offline), so it is validated for what the engine NEEDS, not for what a genuine checkpoint may carry
            # on top (buffers, layer-scale or text-side keys under `visual.`): missing tensors still fail a strict load,
            # extra ones are reported - an unverified key list must not make a whole SAMAudio checkpoint unloadable.
            v_missing, v_unexpected = self.vision_encoder.load_state_dict(vis, strict=False)
            if strict and v_missing:
                raise RuntimeError(f"Missing keys: {['vision_encoder.' + k for k in v_missing]}")
            if v_unexpected:
                warnings.warn(f"vision_encoder: {len(v_unexpected)} checkpoint tensors are not consumed by the PE-Core tower "
                              f"engine (first: {v_unexpected[:3]})")
        with torch.cuda.device(self.device):
            self._lanes = []   # extra stream contexts borrow the weight tensors registered below: rebuilt on demand
            # register everything first: replacing a tensor the engine already holds (a second load_state_dict) marks BOTH
            # weight sets as not finalized, so each set the model has is finalized again afterwards
            if not dit_missing:
                alt = [leaf for leaf, c in hip.ALT16_WEIGHTS.items() if self.alt16_classes & hip.CLS[c]]
                dit = convert_dit(state_dict, self.cfg, self.act_dtype, self.device, alt16_leaves=alt,
                                  f32_classes=self.f32_classes, ktm=self.weight_layout == "ktm")
                self._register(dit)
                if self.x3_classes:
                    self._register(convert_dit_x3(dit, self.cfg.transformer.n_layers, hip.half_dtype(self.precision),
                                                  self.x3_classes))
                # fp32 operand copies exist for the classes that run in fp32 now; set_f32_classes adds a class's later
                # from these references to the checkpoint entries (no copy is made here)
                self._f32_have = self.f32_classes
                self._f32_sd = {k: state_dict[k] for k in F32_SOURCE_KEYS}
                self._has_dit = True
            if not codec_missing:
                codec = convert_codec(state_dict, self.cfg, self.act_dtype, self.device)
                self._register(codec)
                if self.x3_classes & hip.CLS["codec"]:   # the wide convolutions' split twins (8-phase 16-bit launches over K' = 3K)
                    self._register(convert_codec_x3(codec, hip.half_dtype(self.precision)))
                    # ... and the narrow ones' weights split once, for the fp32 kernel that splits its activations on the fly
                    self._register(convert_codec_fly16(codec, hip.half_dtype(self.precision)))
                self._has_codec = True
                if self.codec_decode == "16":
                    self._codec16_tensors = convert_codec(state_dict, self.cfg, hip.half_dtype(self.precision), self.device)
                    self._codec16 = None
                    for lane in self._lanes:
                        lane._codec16 = None
            if self._has_dit and not (dit_missing and codec_missing):
                hip.check(self._lib.samaudio_finalize(self._ctx, 0))
            if self._has_codec and not (dit_missing and codec_missing):
                hip.check(self._lib.samaudio_finalize(self._ctx, 1))
        return missing, unexpected

    def _covers_vision_tower(self, vis: Dict[str, torch.Tensor]) -> bool:
        from .config import PE_VISION_CONFIGS
        from .vision_tower import expected_keys
        pe = PE_VISION_CONFIGS.get(self.cfg.vision_encoder.name)
        return pe is not None and all(("model.visual." + k) in vis for k in expected_keys(pe))

    def _set_precision_options(self, ctx) -> None:
        if hip.storage_precision(self.precision) != "fp32":
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_F32_CLASSES, self.f32_classes))
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_ALT16_CLASSES, self.alt16_classes))
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_PREFETCH_ROWS, self.prefetch_rows))
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_SENTINEL, int(getattr(self, "_sentinel", False))))
        else:
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_QUANT_CLASSES, self.quant_classes))
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_QUANT_FORMAT, self.quant_format))
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_X3_CLASSES, self.x3_classes))

    def set_f32_classes(self, classes) -> None:
        """Switch the exact-fp32 GEMM classes of a 16-bit model (see __init__); takes effect from the next call."""
        self.f32_classes = 0 if hip.storage_precision(self.precision) == "fp32" else hip.class_mask(classes)
        need = self.f32_classes & ~self._f32_have
        if need and self._has_dit:   # the classes switched on for the first time: their "<name>.f32" operand copies
            with torch.cuda.device(self.device):
                self._register(convert_dit_f32(self._f32_sd, self.cfg, self.device, need))
                self._f32_have |= need
                self._lanes = []   # stream lanes borrow the registered tensors: rebuilt on demand
                # binds the new names.  Registering a name the engine already holds (the copies of an earlier checkpoint that
                # a later load_state_dict left behind) marks EVERY weight set as not finalized: each set the model has is
                # finalized again (the codec set was not, and the next separate() failed in the codec)
                hip.check(self._lib.samaudio_finalize(self._ctx, 0))
                if self._has_codec:
                    hip.check(self._lib.samaudio_finalize(self._ctx, 1))
        for ctx in [self._ctx] + [lane._ctx for lane in self._lanes]:
            self._set_precision_options(ctx)

    def set_quantised_classes(self, classes, fmt: str = "bf16") -> None:
        """fp32 models only - measurement aid: the GEMMs of `classes` round both operands to `fmt` ("bf16" | "fp16")
        before multiplying, everything else stays exact (samaudio.h SAMAUDIO_OPT_QUANT_CLASSES)."""
        if hip.storage_precision(self.precision) != "fp32":
            raise ValueError("operand-rounding emulation needs precision='fp32'")
        self.quant_classes, self.quant_format = hip.class_mask(classes), hip.QUANT_FORMATS[fmt]
        for ctx in [self._ctx] + [lane._ctx for lane in self._lanes]:
            self._set_precision_options(ctx)

    def _alt16_weight(self, name: str) -> bool:
        """engine tensor `name` is the weight of a class that reads alt-format (bfloat16) operands in this model"""
        leaf = name.rsplit(".", 1)[-1]
        return name.startswith("L") and leaf in hip.ALT16_WEIGHTS and bool(self.alt16_classes & hip.CLS[hip.ALT16_WEIGHTS[leaf]])

    def _register(self, tensors: Dict[str, torch.Tensor]) -> None:
        for name, t in tensors.items():
            dt = hip.dtype_code(t.dtype, hip.operands_for(self.precision), alt_ok=self._alt16_weight(name))
            if t.data_ptr() % 16:   # a view into a larger buffer: the library needs 16-byte aligned pointers
                t = t.clone()
            self._tensors[name] = t  # keep alive: the library borrows the pointer
            hip.check(self._lib.samaudio_set_tensor(self._ctx, name.encode(), hip.ptr(t), dt, t.dim(),
                                                    hip.shape_array(t.shape)))

    def engine_tensors(self) -> Dict[str, torch.Tensor]:
        return self._tensors

    # ------------------------------------------------------------------ validation aid
    def sentinel(self, on: bool = True) -> None:
        """samaudio.h SAMAUDIO_OPT_SENTINEL: every 16-bit tensor a GEMM of the hot path writes (and the RMSNorm / attention
        outputs that feed GEMMs) is scanned for its largest magnitude and for non-finite values; sentinel_report() reads and
        resets the figures.  An fp16 overflow is then reported with its GEMM class instead of propagating."""
        self._sentinel = bool(on)
        for ctx in [self._ctx] + [lane._ctx for lane in self._lanes]:
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_SENTINEL, int(self._sentinel)))

    def sentinel_report(self) -> Dict[str, Dict[str, float]]:
        """{class: {"absmax": ..., "nonfinite": ...}} over every engine context since the last report (synchronises)."""
        out = {n: {"absmax": 0.0, "nonfinite": 0.0} for n in hip.SENTINEL_NAMES}
        with torch.cuda.device(self.device):
            torch.cuda.synchronize()
            for ctx in [self._ctx] + [lane._ctx for lane in self._lanes]:
                mx, bad = (C.c_float * hip.SENTINEL_SLOTS)(), (C.c_double * hip.SENTINEL_SLOTS)()
                hip.check(self._lib.samaudio_sentinel_read(ctx, mx, bad, hip.current_stream_ptr()))
                for i, n in enumerate(hip.SENTINEL_NAMES):
                    out[n]["absmax"] = max(out[n]["absmax"], float(mx[i]))
                    out[n]["nonfinite"] += float(bad[i])
        return out

    # ------------------------------------------------------------------ measurement (bench.py)
    def profile_begin(self, serial_groups: bool = True) -> None:
        """Bracket every kernel launch with a hipEvent pair on its launch stream until profile_end(), on every engine
        context of the model (the row groups of `streams` > 1 have one each).  `serial_groups`: while profiling, the row
        groups run ONE AFTER THE OTHER on the caller's stream, so that an event pair times a kernel that has the GPU to
        itself - the launches (shapes, tile policy, options) are exactly those of the concurrent run."""
        self._profiling, self._serial_groups = True, bool(serial_groups)
        for ctx in self._profiled_ctxs():
            hip.check(self._lib.samaudio_profile_begin(ctx))

    def _profiled_ctxs(self):
        """every engine context that launches kernels of a step: the model's, its stream lanes', and their 16-bit codec contexts"""
        owners = [self] + list(self._lanes)
        if self.codec_decode == "16" and self._has_codec:
            for own in owners:
                if own._codec16 is None:
                    own._codec16 = _Codec16(self)
        return [o._ctx for o in owners] + [o._codec16._ctx for o in owners if o._codec16 is not None]

    def profile_end(self) -> List[Dict[str, Any]]:
        """[{name, launches, flops, bytes, ms}] per (class, kernel) since profile_begin() (synchronises), summed over the
        contexts.  name = "dit/<kernel>" | "codec/<kernel>" | "prep/<kernel>"; flops / bytes are algorithmic (see
        include/samaudio.h)."""
        merged: Dict[str, Dict[str, Any]] = {}
        for ctx in self._profiled_ctxs():
            buf = (hip.KernelStat * 64)()
            n = C.c_int(0)
            hip.check(self._lib.samaudio_profile_end(ctx, buf, 64, C.byref(n)))
            for i in range(n.value):
                name = buf[i].name.decode()
                row = merged.setdefault(name, dict(name=name, launches=0, flops=0.0, bytes=0.0, ms=0.0))
                row["launches"] += int(buf[i].launches)
                row["flops"] += float(buf[i].flops)
                row["bytes"] += float(buf[i].bytes)
                row["ms"] += float(buf[i].ms)
        self._profiling = self._serial_groups = False
        return list(merged.values())

    # ------------------------------------------------------------------ workspace
    def _ensure_workspace(self, rows: int, frames: int, text_len: int, codec_items: int, samples: int,
                          lane: Optional[_Lane] = None) -> None:
        own = lane if lane is not None else self   # whoever owns the context owns its workspace
        need = self._lib.samaudio_workspace_bytes(own._ctx, rows, frames, max(1, text_len), codec_items, samples)
        if own._workspace is None or own._workspace.numel() < need:
            own._workspace = None
            if os.environ.get("SAMAUDIO_POISON"):
                # test aid: every byte 0xFF = NaN in fp32 and bf16, so a kernel that reads scratch nobody wrote shows up
                # as NaN deterministically instead of depending on what the allocator handed out
                fill = int(os.environ.get("SAMAUDIO_POISON_BYTE", "255"), 0)   # a finite pattern (e.g. 0x3F) shows reads
                own._workspace = torch.full((need + 256,), fill, dtype=torch.uint8, device=self.device)  # that NaN-ignoring ops hide
            else:
                own._workspace = torch.empty(need + 256, dtype=torch.uint8, device=self.device)
        base = own._workspace.data_ptr()
        # contexts never share scratch: the workspaces of the model and of its stream lanes are pairwise disjoint
        for other in [self] + list(self._lanes):
            ws = other._workspace
            if other is not own and ws is not None:
                lo, hi = ws.data_ptr(), ws.data_ptr() + ws.numel()
                if base < hi and lo < base + own._workspace.numel():
                    raise RuntimeError("engine contexts were handed overlapping workspaces")
        aligned = (base + 255) // 256 * 256
        hip.check(self._lib.samaudio_set_workspace(own._ctx, C.c_void_p(aligned),
                                                   own._workspace.numel() - (aligned - base)))

    def _codec_chunk(self, items: int) -> int:
        return min(items, int(os.environ.get("SAMAUDIO_CODEC_CHUNK", "16")))

    # ------------------------------------------------------------------ codec (reference codec.py)
    def _pad_to_hop(self, wavs: torch.Tensor) -> torch.Tensor:
        hop = self.cfg.audio_codec.hop_length
        rem = wavs.size(-1) % hop
        return wavs if rem == 0 else torch.nn.functional.pad(wavs, (0, hop - rem), mode="reflect")

    def encode_audio(self, audios: torch.Tensor) -> torch.Tensor:
        """audios [B,1,Tw] -> mean latent, channels-last [B, T, codebook_dim] (reference codec.py:65-78;
        the reference returns [B, C, T] and transposes at model.py:183)."""
        if not self._has_codec:
            raise RuntimeError("audio_codec weights are not loaded")
        wav = self._pad_to_hop(audios.to(self.device, torch.float32)).squeeze(1).contiguous()
        items, samples = wav.shape
        frames = samples // self.cfg.audio_codec.hop_length
        z = torch.empty(items, frames, self.cfg.audio_codec.codebook_dim, device=self.device)
        with torch.cuda.device(self.device):
            self._ensure_workspace(0, 0, 0, self._codec_chunk(items), samples)
            hip.check(self._lib.samaudio_codec_encode(self._ctx, hip.ptr(wav), items, samples, hip.ptr(z),
                                                      hip.current_stream_ptr()))
        return z

    def decode_audio(self, latents: torch.Tensor, lane: Optional[_Lane] = None,
                     out: Optional[torch.Tensor] = None, pairs: bool = False) -> torch.Tensor:
        """latents channels-last [N, T, codebook_dim] -> [N, T*hop] (reference codec.py:86-89).  `lane`: run on that
        lane's context (and on the current stream); `out`: a contiguous [N, T*hop] fp32 destination.
        `pairs`: `latents` is the ODE state [rows, T, 2*codebook_dim]; waveform 2b = the first codebook_dim channels of row
        block b (target), 2b + 1 the rest (residual) - reference model.py:291-295, gathered inside the engine instead of a
        transposed copy."""
        if not self._has_codec:
            raise RuntimeError("audio_codec weights are not loaded")
        lat = latents.to(self.device, torch.float32).contiguous()
        items, frames, _ = lat.shape
        if pairs:
            assert lat.shape[2] == 2 * self.cfg.audio_codec.codebook_dim
            items *= 2
        samples = frames * self.cfg.audio_codec.hop_length
        wav = out if out is not None else torch.empty(items, samples, device=self.device)
        assert wav.shape == (items, samples) and wav.is_contiguous() and wav.dtype == torch.float32
        with torch.cuda.device(self.device):
            self._ensure_workspace(0, 0, 0, self._codec_chunk(items), samples, lane=lane)
            ctx = self._ctx if lane is None else lane._ctx
            if self.codec_decode == "16":
                # the 16-bit codec context of this owner, on the owner's workspace (sized for the fp32 context's codec pass, which
                # needs more; stream order keeps the two contexts' uses of it apart)
                own = lane if lane is not None else self
                if own._codec16 is None:
                    own._codec16 = _Codec16(self)
                ctx = own._codec16._ctx
                base = own._workspace.data_ptr()
                aligned = (base + 255) // 256 * 256
                need = self._lib.samaudio_workspace_bytes(ctx, 0, 0, 1, self._codec_chunk(items), samples)
                assert own._workspace.numel() - (aligned - base) >= need
                hip.check(self._lib.samaudio_set_workspace(ctx, C.c_void_p(aligned), own._workspace.numel() - (aligned - base)))
            if pairs:
                hip.check(self._lib.samaudio_codec_decode_pairs(ctx, hip.ptr(lat), items // 2, frames, hip.ptr(wav),
                                                                hip.current_stream_ptr()))
            else:
                hip.check(self._lib.samaudio_codec_decode(ctx, hip.ptr(lat), items, frames, hip.ptr(wav),
                                                          hip.current_stream_ptr()))
        return wav

    # ------------------------------------------------------------------ DiT
    def _prepare(self, audio_features, text_features, text_mask, masked_video_features, anchor_ids,
                 anchor_alignment, audio_pad_mask, lane: Optional[_Lane] = None, anchors_validated: bool = False,
                 candidates: int = 1, latent: bool = False) -> None:
        """`latent`: `audio_features` is the codec latent z [B, T, codebook_dim] and every conditioning tensor holds B clips, each
        serving `candidates` consecutive rows of the solve (samaudio_prepare_latent: (z | z) and the sample-major repeat of
        reference model.py:182-184,193-203 happen inside the engine, nothing is concatenated or repeated here)."""
        dev = self.device
        own = lane if lane is not None else self
        feats = audio_features.to(dev, torch.float32).contiguous()
        rows, frames, _ = feats.shape
        text = tmask = video = ids = align = pad = None
        text_len = 1
        if text_features is not None:
            text = text_features.to(dev, torch.float32).contiguous()
            assert text.size(0) == rows, "text_features batch mismatch"
            text_len = text.size(1)
            if text_mask is not None:
                tmask = text_mask.to(dev).to(torch.uint8).contiguous()
        if masked_video_features is not None:  # reference layout [B, C, T] -> channels-last
            video = masked_video_features.to(dev, torch.float32).transpose(1, 2).contiguous()
            assert video.shape[:2] == (rows, frames), "masked_video_features must be [B, C, T]"
        n_ids = 0
        if anchor_ids is not None:
            ids = anchor_ids.to(dev, torch.long).contiguous()
            align = anchor_alignment.to(dev, torch.long).contiguous()
            n_ids = ids.size(1)
            # the reference's gather / nn.Embedding raise on out-of-range indices (model.py:61).  Tensors a Batch built on the
            # host are in range by construction (processor.Batch.process_anchors checks them there); anything else - forward()
            # called with hand-made tensors - is checked here, at the price of blocking device -> host reads
            if not anchors_validated:
                lo, hi = int(align.min()), int(align.max())
                if lo < 0 or hi >= n_ids:
                    raise IndexError(f"anchor_alignment values must be in [0, {n_ids}): found [{lo}, {hi}]")
                lo, hi = int(ids.min()), int(ids.max())
                if lo < 0 or hi > self.cfg.num_anchors:
                    raise IndexError(f"anchor_ids must be in [0, {self.cfg.num_anchors}]: found [{lo}, {hi}]")
        if audio_pad_mask is not None:
            pad = audio_pad_mask.to(dev).to(torch.uint8).contiguous()
        own._live = (feats, text, tmask, video, ids, align, pad)
        if latent:
            self._ensure_workspace(rows * candidates, frames, text_len, 0, 0, lane)
            hip.check(self._lib.samaudio_prepare_latent(
                own._ctx, rows * candidates, frames, text_len, candidates, hip.ptr(feats), hip.ptr(text), hip.ptr(tmask),
                hip.ptr(video), hip.ptr(ids), n_ids, hip.ptr(align), hip.ptr(pad), hip.current_stream_ptr()))
            return
        assert candidates == 1
        self._ensure_workspace(rows, frames, text_len, 0, 0, lane)
        hip.check(self._lib.samaudio_prepare(
            own._ctx, rows, frames, text_len, hip.ptr(feats), hip.ptr(text), hip.ptr(tmask), hip.ptr(video),
            hip.ptr(ids), n_ids, hip.ptr(align), hip.ptr(pad), hip.current_stream_ptr()))

    def forward(self, noisy_audio: torch.Tensor, audio_features: torch.Tensor, text_features: torch.Tensor,
                time: torch.Tensor, masked_video_features: Optional[torch.Tensor] = None,
                text_mask: Optional[torch.Tensor] = None, anchor_ids: Optional[torch.Tensor] = None,
                anchor_alignment: Optional[torch.Tensor] = None,
                audio_pad_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One ODE function evaluation; same signature as reference model.py:130-141."""
        if not self._has_dit:
            raise RuntimeError("transformer weights are not loaded")
        with torch.cuda.device(self.device):
            self._prepare(audio_features, text_features, text_mask, masked_video_features, anchor_ids,
                          anchor_alignment, audio_pad_mask)
            noisy = noisy_audio.to(self.device, torch.float32).contiguous()
            t = time.to(self.device, torch.float32).reshape(-1).contiguous()
            out = torch.empty_like(noisy)
            hip.check(self._lib.samaudio_forward(self._ctx, hip.ptr(noisy), hip.ptr(t), t.numel(), hip.ptr(out),
                                                 hip.current_stream_ptr()))
        return out

    __call__ = forward

    def solve(self, noise: torch.Tensor, ode_opt: Dict[str, Any] = DFLT_ODE_OPT) -> torch.Tensor:
        """Integrate the flow ODE from `noise` with the conditioning of the last `_prepare` call."""
        method, grid = ode_grid(ode_opt)
        state = noise.to(self.device, torch.float32).clone().contiguous()
        g = (C.c_float * len(grid))(*grid)
        with torch.cuda.device(self.device):
            hip.check(self._lib.samaudio_ode_solve(self._ctx, hip.ptr(state), method, g, len(grid),
                                                   hip.current_stream_ptr()))
        return state

    def _apply_options(self, groups: int) -> None:
        split = int(self.tail_split if self.tail_split is not None else groups == 1)
        for ctx in [self._ctx] + [lane._ctx for lane in self._lanes]:
            hip.check(self._lib.samaudio_set_option(ctx, hip.OPT_TAIL_SPLIT, split))

    def _solve_concurrent(self, noise: torch.Tensor, ode_opt: Dict[str, Any], cond: List[Optional[torch.Tensor]],
                          groups: int, decode: bool = False, anchors_validated: bool = False, candidates: int = 1,
                          latent: bool = False):
        """prepare + ODE solve (+ DAC-VAE decode of target and residual when `decode`) of `groups` contiguous row groups,
        each on its own engine context and HIP stream, driven by one host thread per group (the C calls release the GIL).
        Rows are independent, so the result equals the single-stream one bit for bit.  Returns the
        latent state, and with `decode` also the waveforms [rows, 2, samples] (model.py:291-295)."""
        import threading
        from .dist import shard_range
        method, grid = ode_grid(ode_opt)
        g = (C.c_float * len(grid))(*grid)
        state = noise.to(self.device, torch.float32).clone().contiguous()
        rows, frames, C2 = state.shape
        wavs = (torch.empty(rows * 2, frames * self.cfg.audio_codec.hop_length, device=self.device)
                if decode else None)
        while len(self._lanes) < groups - 1:
            self._lanes.append(_Lane(self))
            if self._profiling:
                hip.check(self._lib.samaudio_profile_begin(self._lanes[-1]._ctx))
                if self.codec_decode == "16":
                    self._lanes[-1]._codec16 = _Codec16(self)
                    hip.check(self._lib.samaudio_profile_begin(self._lanes[-1]._codec16._ctx))
        self._apply_options(groups)
        main = torch.cuda.current_stream(self.device)
        errors: List[BaseException] = []

        def work(i: int, lane: Optional[_Lane]):
            try:
                # contiguous groups of CLIPS (all candidates of a clip stay in one group: SURVEY.md section 8e)
                cr = shard_range(rows // candidates, i, groups)
                rr = range(cr.start * candidates, cr.stop * candidates)
                sl = slice(rr.start, rr.stop)
                csl = slice(cr.start, cr.stop) if latent else sl   # `latent`: the conditioning is per clip
                part = [None if c is None else c[csl] for c in cond]
                stream = main if (lane is None or self._serial_groups) else lane.stream
                if stream is not main:
                    # tensors allocated on the caller's stream and used on the lane's: tell the caching allocator, so that
                    # none of their blocks is handed out again before the lane's work on them has finished (the explicit
                    # wait_stream pair below already orders it for the tensors this frame holds; this covers the views
                    # `_prepare` keeps in `lane._live` beyond this call)
                    for t in [state, wavs] + list(cond):
                        if t is not None and t.is_cuda:
                            t.record_stream(stream)
                with torch.inference_mode(), torch.cuda.device(self.device), torch.cuda.stream(stream):
                    self._prepare(*part, lane=lane, anchors_validated=anchors_validated, candidates=candidates, latent=latent)
                    ctx = self._ctx if lane is None else lane._ctx
                    hip.check(self._lib.samaudio_ode_solve(ctx, hip.ptr(state[sl]), method, g, len(grid),
                                                           hip.current_stream_ptr()))
                    if decode:   # waveforms (2b, 2b+1) = (target, residual) of row b, gathered from the state inside the engine
                        self.decode_audio(state[sl], lane=lane, out=wavs[2 * rr.start: 2 * rr.stop], pairs=True)
            except BaseException as exc:  # re-raised on the caller's thread
                errors.append(exc)

        if self._serial_groups:   # profiling: the same launches, one group after the other on the caller's stream
            for i in range(groups):
                work(i, None if i == 0 else self._lanes[i - 1])
        else:
            for lane in self._lanes[:groups - 1]:
                lane.stream.wait_stream(main)
            threads = [threading.Thread(target=work, args=(i, None if i == 0 else self._lanes[i - 1])) for i in range(groups)]
            for th in threads:
                th.start()
            for th in threads:
                th.join()
            for lane in self._lanes[:groups - 1]:
                main.wait_stream(lane.stream)
        if errors:
            raise errors[0]
        return (state, wavs.view(rows, 2, -1)) if decode else state

    # ------------------------------------------------------------------ separate()
    def _text(self, batch: Batch):
        if batch.text_features is not None:
            feats = batch.text_features
            mask = batch.text_mask if batch.text_mask is not None else torch.ones(feats.shape[:2], dtype=torch.bool)
            return feats, mask
        if self.text_encoder is None:
            raise RuntimeError(
                "no text encoder is attached (t5-base cannot be downloaded offline): pass "
                "text_features/text_mask to the processor or set model.text_encoder")
        return self.text_encoder(batch.descriptions)

    @staticmethod
    def _repeat(x: Optional[torch.Tensor], candidates: int):
        """Sample-major repeat, reference model.py:193-203."""
        if x is None or candidates == 1:
            return x
        return x.repeat_interleave(candidates, dim=0)

    @torch.inference_mode()
    def separate(self, batch: Batch, noise: Optional[torch.Tensor] = None,
                 ode_opt: Dict[str, Any] = DFLT_ODE_OPT, reranking_candidates: int = 1,
                 predict_spans: bool = False) -> SeparationResult:
        """Reference model.py:247-338.  `reranking_candidates > 1` draws that many ODE solutions per clip and lets
        `visual_ranker` / `text_ranker` pick one (model.py:306-330); `predict_spans` runs `span_predictor` first
        (model.py:259-268; mind quirk Q13 below)."""
        if not (self._has_dit and self._has_codec):
            raise RuntimeError("load_state_dict() first")
        cand = int(reranking_candidates)
        with torch.cuda.device(self.device):
            # Nothing of torch's own arithmetic runs between the first and the last kernel of a step: the noise is drawn first
            # (model.py:274-275: `torch.randn_like` on the model device), features = (z | z) (model.py:182-184), the sample-major
            # repeat for the candidates (model.py:193-203) and the (target, residual) split of the state (model.py:291-295) are
            # index arithmetic inside the engine (samaudio_prepare_latent / samaudio_codec_decode_pairs).
            hop = self.cfg.audio_codec.hop_length
            B, T = batch.audios.size(0), -(-batch.audios.size(-1) // hop)
            C2 = self.cfg.transformer.out_channels
            if noise is None:
                noise = torch.randn(B * cand, T, C2, device=self.device)         # model.py:274-275
            assert tuple(noise.shape) == (B * cand, T, C2), "noise must be [B*candidates, T, 256]"
            z = self.encode_audio(batch.audios)                                  # [B, T, 128]
            assert z.shape[:2] == (B, T) and 2 * z.size(2) == C2
            text, text_mask = self._text(batch)
            video = None
            if batch.masked_video is not None:                                   # model.py:186-191
                if self.vision_encoder is None:
                    raise NotImplementedError(
                        "visual prompting needs a vision encoder: set model.vision_encoder to a callable "
                        "list[video [T,3,H,W]] -> features [B, T, vision_encoder.dim] (PE-Core tower, SURVEY.md section 8 f3)")
                video = self.vision_encoder(batch.masked_video).transpose(1, 2)  # reference layout [B, C, T]
            # forward args are assembled BEFORE the span predictor runs (reference model.py:257 vs :259-268), and
            # process_anchors rebinds new tensors, so in the reference snapshot predicted spans never reach the ODE
            # (quirk Q13).  fix_span_order=True opts into the evidently intended order.
            anchor_ids, anchor_alignment = batch.anchor_ids, batch.anchor_alignment
            # host-built anchors, range-checked against THIS model's anchor vocabulary: no device-side range check
            validated = getattr(batch, "anchor_vocab_validated", 0) == self.cfg.num_anchors + 1
            if predict_spans and batch.anchors is None:
                if self.span_predictor is None:
                    warnings.warn("predict_spans=True ignored: no span predictor attached (model.span_predictor)")
                else:
                    batch = self.predict_spans(batch, z, batch.audio_pad_mask)   # model.py:259-268 (reads channels [0, 128))
                    if self.fix_span_order:
                        anchor_ids, anchor_alignment = batch.anchor_ids, batch.anchor_alignment
            cond = [z, text, text_mask, video, anchor_ids, anchor_alignment, batch.audio_pad_mask]   # per clip: B items each
            groups = min(self.streams, B)
            wavs = None
            if groups > 1:
                cond = [None if c is None else c.to(self.device) for c in cond]
                # each group also decodes its own rows on its stream: the codec's HBM-bound convolutions of one group run
                # beside the other group's kernels instead of after both solves
                latent, wavs = self._solve_concurrent(noise, ode_opt, cond, groups, decode=True, anchors_validated=validated,
                                                      candidates=cand, latent=True)
            else:
                self._apply_options(1)
                self._prepare(*cond, anchors_validated=validated, candidates=cand, latent=True)
                latent = self.solve(noise, ode_opt)                              # states[-1], [Bc, T, 256]
            self.last_latent = latent
            # [Bc, T, 2C] -> waveforms (2b, 2b+1) = (target, residual) of row b (model.py:291-295)
            Bc = latent.size(0)
            if wavs is None:
                wavs = self.decode_audio(latent, pairs=True).view(Bc, 2, -1)
            # codec.py:91-97, from the batch's host copy of the frame counts: slicing by device scalars would block on the GPU
            sizes = [n * hop for n in (getattr(batch, "sizes_host", None) or [int(v) for v in batch.sizes.tolist()])]
            target = self.unbatch(wavs[:, 0].view(B, cand, -1), sizes)
            residual = self.unbatch(wavs[:, 1].view(B, cand, -1), sizes)
            if cand == 1:   # nothing to pick: views, no index kernels, no device -> host reads
                return SeparationResult(target=[w[0] for w in target], residual=[w[0] for w in residual], noise=noise)
            idxs = self._rerank(batch, target, sizes, cand)                      # model.py:306-330
            return SeparationResult(
                target=[w[i] for w, i in zip(target, idxs)],
                residual=[w[i] for w, i in zip(residual, idxs)],
                noise=noise)

    def _rerank(self, batch: Batch, target_wavs: List[torch.Tensor], sizes: List[int], cand: int) -> torch.Tensor:
        """Candidate selection, reference model.py:306-330: visual ranker if a masked video came with the batch, else
        the text ranker, else candidate 0; `idxs = scores.argmax(dim=1)`."""
        B = len(target_wavs)
        sr = self.cfg.audio_codec.sample_rate
        if cand > 1 and self.text_ranker is None and (self.visual_ranker is None or batch.masked_video is None):
            warnings.warn(f"reranking_candidates={cand} but no applicable ranker is attached (model.text_ranker / "
                          "model.visual_ranker): candidate 0 is returned for every clip")
        if cand > 1 and batch.masked_video is not None and self.visual_ranker is not None:
            scores = self.visual_ranker(extracted_audio=target_wavs, videos=batch.masked_video, sample_rate=sr)
            return scores.argmax(dim=1)
        if cand > 1 and self.text_ranker is not None:
            input_audio = [audio[:, : int(size)].expand(cand, -1) for audio, size in zip(batch.audios, sizes)]
            scores = self.text_ranker(extracted_audio=target_wavs, input_audio=input_audio,
                                      descriptions=batch.descriptions, sample_rate=sr)
            return scores.argmax(dim=1)
        return torch.zeros(B, dtype=torch.long, device=self.device)

    def predict_spans(self, batch: Batch, audio_features: torch.Tensor, audio_pad_mask: torch.Tensor) -> Batch:
        """reference model.py:231-245: PE-A-Frame on the first 128 feature channels (the codec mean latent) and the
        descriptions -> spans -> "+" anchors -> batch.process_anchors."""
        if self.span_predictor_transform is None:
            raise RuntimeError("predict_spans needs model.span_predictor_transform (descriptions -> the span predictor's "
                               "text inputs: input_ids / attention_mask, or text_pooled)")
        inputs = self.span_predictor_transform(text=batch.descriptions)
        inputs = {k: v for k, v in dict(inputs).items() if k in ("input_ids", "attention_mask", "text_pooled")}
        half = self.cfg.audio_codec.codebook_dim
        output = self.span_predictor(input_features=audio_features[:, :, :half].contiguous(),
                                     padding_mask=audio_pad_mask, return_spans=True, **inputs)
        anchors = [[("+", float(s), float(e)) for s, e in spans] for spans in output.spans]
        batch.process_anchors(anchors)
        return batch

    def attach_rankers(self, **kwargs) -> None:
        """Build the rankers named in the config (reference model.py:94-95) when they point at LOCAL checkpoint
        directories; hub ids cannot be resolved offline and are left unattached (a warning says so)."""
        from .ranking import create_ranker
        for name in ("visual_ranker", "text_ranker"):
            rc = getattr(self.cfg, name)
            if rc is None or getattr(self, name) is not None:
                continue
            path = getattr(rc, "checkpoint_or_model_id", None)
            if path is not None and not os.path.isdir(path):
                warnings.warn(f"{name}: {path!r} is not a local directory (no hub access offline); not attached")
                continue
            try:
                setattr(self, name, create_ranker(rc, device=str(self.device) if self.device else None, **kwargs))
            except NotImplementedError as exc:  # CLAP / ImageBind / ... wrap third-party models this build does not ship
                warnings.warn(f"{name} not attached: {exc}")

    def unbatch(self, wavs: torch.Tensor, sizes, time_dim: int = -1):
        """reference model.py:340-344"""
        return [row.narrow(dim=time_dim, start=0, length=int(size)) for row, size in zip(wavs, sizes)]
