"""`T5TextEncoder` - host wrapper with the interface of the reference's text encoder
(reference sam_audio/model/text_encoder.py:11-37).

The prompt encoder runs once per `separate()` call on a handful of tokens (pad-to-longest, Lt ~ 2-16 for
noun-phrase prompts): at 32 prompts it is ~40 GFLOP against the 1.4 PFLOP of the ODE, so it stays on
PyTorch-ROCm (`transformers.T5EncoderModel`, rocBLAS GEMMs) exactly as SURVEY.md prescribes; the HIP path
starts at the `[B, Lt, 768]` features it returns.  There is no network in this build's environment, so the
model / tokenizer are taken from a local directory (`cfg.name` may be a path) or from the local HF cache
(`local_files_only=True`); a missing checkpoint raises instead of silently producing random features.
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
                 device=None, dtype: Optional[torch.dtype] = None):
        self.cfg = cfg or T5EncoderConfig()
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
        if device is not None or dtype is not None:
            self.to(device=device, dtype=dtype)
        width = getattr(getattr(self.model, "config", None), "d_model", None)
        if width is not None and width != self.cfg.dim:
            raise ValueError(f"text encoder width {width} != T5EncoderConfig.dim {self.cfg.dim}")

    def to(self, device=None, dtype=None):
        self.model = self.model.to(device=device, dtype=dtype)
        return self

    @property
    def device(self) -> torch.device:
        return next(self.model.parameters()).device

    @torch.inference_mode()
    def forward(self, texts: List[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        # reference text_encoder.py:19-37
        encoded = self.tokenizer(texts, truncation=True, max_length=self.max_length, padding=self.pad_mode,
                                 return_tensors="pt")
        device = self.device
        input_ids = encoded["input_ids"].to(device)
        attention_mask = encoded["attention_mask"].to(device)
        res = self.model(input_ids=input_ids, attention_mask=attention_mask,
                         output_hidden_states=True)["last_hidden_state"]
        return res, attention_mask.bool()

    __call__ = forward
