"""TEST INFRASTRUCTURE: the `transformers` modules executed by PyTorch, as checkers of the HIP text stacks and as the text towers
of the CPU dry runs (tests/conftest.py, launcher emulation: it carries no text-tower kernels).  The product has no PyTorch
execution path for these (sam_audio_amd/text_encoder.py, sam_audio_amd/judge.py _TextTower): this file used to live there as
`backend="torch"`."""
from typing import Optional

import torch


class TorchTextTower:
    """Same interface as sam_audio_amd.judge._TextTower, forward = the ModernBertModel module itself."""

    def __init__(self, module):
        self.module = module
        self._device = None

    def place(self, device) -> None:
        self._device = torch.device(device)
        self.module = self.module.to(self._device).eval()

    def hidden(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor], nth: Optional[int],
               last_prenorm: bool = True) -> torch.Tensor:
        layers = self.module.config.num_hidden_layers
        grabbed = []
        hook = None
        if nth == layers and last_prenorm:   # the input of final_norm IS the transformers-4.x hidden_states[layers]
            hook = self.module.final_norm.register_forward_pre_hook(lambda mod, args: grabbed.append(args[0]))
        try:
            out = self.module(input_ids=input_ids.to(self._device),
                              attention_mask=None if attention_mask is None else attention_mask.to(self._device),
                              output_hidden_states=nth is not None and nth != layers)
        finally:
            if hook is not None:
                hook.remove()
        if nth is None or (nth == layers and not last_prenorm):
            return out.last_hidden_state
        return grabbed[0] if nth == layers else out.hidden_states[nth]


def t5_features(module, tokenizer, texts, max_length=512, pad_mode="longest"):
    """reference text_encoder.py:19-37 on the `transformers.T5EncoderModel` module itself"""
    enc = tokenizer(texts, truncation=True, max_length=max_length, padding=pad_mode, return_tensors="pt")
    with torch.inference_mode():
        res = module(input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], output_hidden_states=True)
    return res["last_hidden_state"], enc["attention_mask"].bool()
