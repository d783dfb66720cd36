"""`T5TextEncoder` - host wrapper with the interface of the reference's text encoder
(reference sam_audio/model/text_encoder.py:11-37).

The prompt encoder runs once per `separate()` call on a handful of tokens (pad-to-longest, Lt ~ 2-16 for
noun-phrase prompts): at 32 prompts it is ~40 GFLOP against the 1.4 PFLOP of the ODE.  Tokenisation is the Hugging
Face tokenizer's, as in the reference.  The encoder stack itself runs on the HIP library and nowhere else:
`sam_audio_amd.t5_encoder.T5EncoderHIP` re-lays the `T5EncoderModel` weights out once and the forward goes through
`samaudio_t5_*` (csrc/t5.hip) in `precision` ("fp32" by default: the stack is tiny next to the ODE and fp32 keeps the
features at the reference's own precision).  The `transformers` module is only the CONTAINER the weights arrive in; it is
never executed by the product (the parity tests run it as their checker: tests/test_t5_gpu.py).  T5 variants the HIP stack
does not build (gated feed-forward) are refused when the weights move to the GPU.

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

    backend = "hip"   # the only one: there is no PyTorch execution path in the product

    def __init__(self, cfg: Optional[T5EncoderConfig] = None, model=None, tokenizer: Optional[Callable] = None,
                 device=None, precision: str = "fp32"):
        self.cfg = cfg or T5EncoderConfig()
        self.precision = precision
        self._hip = None                  # T5EncoderHIP, built when the weights move to the GPU
        self._device = None
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
        self.model = model.eval()         # weight container only
        self.tokenizer = tokenizer
        width = getattr(getattr(self.model, "config", None), "d_model", None)
        if width is not None and width != self.cfg.dim:
            raise ValueError(f"text encoder width {width} != T5EncoderConfig.dim {self.cfg.dim}")
        if device is not None:
            self.to(device)

    def to(self, device):
        """Build the HIP stack's (re-laid out) copy of the weights on a ROCm GPU; any other device is refused."""
        from . import hip
        from .t5_encoder import T5Dims, T5EncoderHIP
        dev = torch.device(device)
        hip.require_gpu(dev, "T5TextEncoder")   # raises "... needs a ROCm GPU: there is no CPU fallback"
        dims = T5Dims.from_hf(self.model.config, max_len=self.max_length)
        enc = T5EncoderHIP(dims, precision=self.precision, device=str(dev))
        enc.load_state_dict(self.model.state_dict())
        self._hip, self._device = enc, dev
        return self

    @property
    def device(self) -> Optional[torch.device]:
        return self._device

    @torch.inference_mode()
    def forward(self, texts: List[str]) -> Tuple[torch.Tensor, torch.Tensor]:
        # reference text_encoder.py:19-37
        if self._hip is None:
            from . import hip
            raise hip.SamAudioHipError("T5TextEncoder needs a ROCm GPU: call .to('cuda') first; there is no CPU fallback")
        encoded = self.tokenizer(texts, truncation=True, max_length=self.max_length, padding=self.pad_mode,
                                 return_tensors="pt")
        ids_cpu = encoded["input_ids"]
        if ids_cpu.numel() and (int(ids_cpu.min()) < 0 or int(ids_cpu.max()) >= self._hip.dims.vocab_size):
            raise IndexError(f"token id outside [0, {self._hip.dims.vocab_size})")   # nn.Embedding raises IndexError too
        # the host -> device copies are issued from pinned staging tensors and do not wait for the GPU: inside separate() the DAC
        # encode is still running on the stream while the prompt is tokenised and the T5 stack is queued behind it
        def up(t):
            return t.pin_memory().to(self._device, non_blocking=True) if self._device.type == "cuda" else t.to(self._device)

        input_ids, attention_mask = up(ids_cpu), up(encoded["attention_mask"])
        return self._hip.encode(input_ids, attention_mask, ids_checked=True), attention_mask.bool()

    __call__ = forward
