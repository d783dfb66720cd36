"""`T5TextEncoder` - host wrapper with the interface of the reference's text encoder
(reference sam_audio/model/text_encoder.py:11-37).

The prompt encoder runs once per `separate()` call on a handful of tokens (pad-to-longest, Lt ~ 2-16 for
noun-phrase prompts): at 32 prompts it is ~40 GFLOP against the 1.4 PFLOP of the ODE.  Tokenisation is the Hugging
Face tokenizer's, as in the reference.  The encoder stack itself has two backends:

* `backend="hip"` (default on a GPU): `sam_audio_amd.t5_encoder.T5EncoderHIP` - the `T5EncoderModel` weights are
  re-laid out once and the forward runs on the HIP library (`samaudio_t5_*`, csrc/t5.hip) in `precision`
  ("fp32" by default: the stack is tiny next to the ODE and fp32 keeps the features at the reference's own precision);
* `backend="torch"`: the `transformers.T5EncoderModel` module on PyTorch-ROCm (rocBLAS GEMMs), what SURVEY.md
  prescribed for round 1 - kept for T5 variants the HIP stack does not build (gated feed-forward) and as a second
  opinion.  It is chosen explicitly, never as a silent fallback.

There is no network in this build's environment, so the model / tokenizer are taken from a local directory
(`cfg.name` may be a path) or from the local HF cache (`local_files_only=True`); a missing checkpoint raises instead
of silently producing random features.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Tuple

import torch

from .config import T5EncoderConfig


class T5TextEncoder:
    """texts -> (last_hidden_state [B, Lt, dim], attention_mask.bool() [B, Lt]).

    `model` / `tokenizer` may be injected (tests use a random-initialised T5 config and a whitespace tokenizer;
    a deployment passes objects it already holds).  Otherwise they are loaded from `cfg.name`.
    """

    def __init__(self, cfg: Optional[T5EncoderConfig] = None, model=None, tokenizer: Optional[Callable] = None,
                 device=None, dtype: Optional[torch.dtype] = None, backend: Optional[str] = None,
                 precision: str = "fp32"):
        self.cfg = cfg or T5EncoderConfig()
        if backend not in (None, "hip", "torch"):
            raise ValueError("backend must be 'hip' or 'torch'")
        self._requested = backend         # what the caller asked for; None = automatic
        self.backend = backend            # the RESOLVED backend: re-decided by every .to(device) when automatic
        self.precision = precision
        self._hip = None                  # T5EncoderHIP, built when the weights move to the GPU
        self.pad_mode = self.cfg.pad_mode
        self.max_length = self.cfg.max_length
        if model is None or tokenizer is None:
            import transformers
            src = self.cfg.name
            local = os.path.isdir(src)
            try:
                if model is None:
                    model = transformers.T5EncoderModel.from_pretrained(src, local_files_only=True)
                if tokenizer is None:
                    tokenizer = transformers.AutoTokenizer.from_pretrained(src, local_files_only=True)
            except Exception as exc:  # OSError from the hub layer, ValueError from sentencepiece, ...
                where = "directory" if local else "local Hugging Face cache entry"
                raise FileNotFoundError(
                    f"T5 text encoder {src!r}: no usable {where} (this build has no network access). "
                    "Point SAMAudioConfig.text_encoder.name at a directory holding the t5-base model + tokenizer, "
                    "or pass text_features/text_mask to the processor.") from exc
        self.model = model.eval()
        self.tokenizer = tokenizer
        width = getattr(getattr(self.model, "config", None), "d_model", None)
        if width is not None and width != self.cfg.dim:
            raise ValueError(f"text encoder width {width} != T5EncoderConfig.dim {self.cfg.dim}")
        if device is not None or dtype is not None:
            self.to(device=device, dtype=dtype)

    def to(self, device=None, dtype=None):
        """Automatic backend (backend=None at construction): "hip" whenever the encoder is on a GPU, "torch" on the CPU,
        re-resolved on EVERY device move (a CPU stop-over does not pin the torch module for a later .to('cuda')).  An
        explicit backend="hip" cannot live on the CPU: the move is refused."""
        dev = torch.device(device) if device is not None else None
        if dev is None:                   # dtype-only request: the device, hence the automatic choice, is unchanged
            backend = self.backend or self._requested or "torch"
        else:
            backend = self._requested or ("hip" if dev.type == "cuda" else "torch")
        if backend == "hip" and dev is not None:
            from . import hip
            # an explicit backend="hip" cannot live on the CPU: raises "... needs a ROCm GPU: there is no CPU fallback"
            hip.require_gpu(dev, "T5TextEncoder(backend='hip') (construct it with backend=None or 'torch' for CPU use)")
        if backend == "hip" and dev is not None:
            # the HIP stack takes its own (re-laid out) copy of the weights; the torch module stays where it is
            from .t5_encoder import T5Dims, T5EncoderHIP
            dims = T5Dims.from_hf(self.model.config, max_len=self.max_length)
            enc = T5EncoderHIP(dims, precision=self.precision, device=str(dev))
            enc.load_state_dict(self.model.state_dict())
            self._hip, self._device = enc, dev
        elif backend == "hip":
            pass                          # dtype-only request: the HIP stack's operand format is `precision`
        else:
            self.model = self.model.to(device=device, dtype=dtype)
            self._hip = None              # a stale device copy must not outlive the move
            self._device = None
        self.backend = backend
        return self

    @property
    def device(self) -> torch.device:
        if self._hip is not None:
            return self._device
        return next(self.model.parameters()).device

    @torch.inference_mode()
    def forward(self, texts: List[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        # reference text_encoder.py:19-37
        encoded = self.tokenizer(texts, truncation=True, max_length=self.max_length, padding=self.pad_mode,
                                 return_tensors="pt")
        device = self.device
        input_ids = encoded["input_ids"].to(device)
        attention_mask = encoded["attention_mask"].to(device)
        if self.backend == "hip":
            if self._hip is None:
                from . import hip
                raise hip.SamAudioHipError("T5TextEncoder(backend='hip') needs a ROCm GPU: call .to('cuda') first; "
                                           "there is no CPU fallback")
            res = self._hip(input_ids, attention_mask)
        else:
            res = self.model(input_ids=input_ids, attention_mask=attention_mask,
                             output_hidden_states=True)["last_hidden_state"]
        return res, attention_mask.bool()

    __call__ = forward
