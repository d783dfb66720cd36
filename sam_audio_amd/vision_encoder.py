"""`PerceptionEncoder` - host wrapper with the interface of the reference's visual-prompt encoder
(reference sam_audio/model/vision_encoder.py:40-113; SURVEY.md section 8 row a4 / "next" row f3).

This wrapper owns what the reference file owns - resize, scaling, normalisation, chunking by `batch_size`, time
padding.  The tower is `pe.CLIP.from_config(cfg.name)` in the reference (un-vendored perception_models); here it is
`sam_audio_amd.vision_tower.PEVisionTower`, the same network on the HIP library (built by default for the config
names in `config.PE_VISION_CONFIGS`), or any injected callable `encode_image(frames [N,3,S,S] float, normalize=bool)
-> [N, dim]`.  `SAMAudio.vision_encoder = PerceptionEncoder(cfg.vision_encoder, device=...)` + `load_state_dict`
enables `separate()` with `masked_videos`.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch

from .config import PerceptionEncoderConfig

_MODES = {"NEAREST": "nearest", "BILINEAR": "bilinear", "BICUBIC": "bicubic"}


class PerceptionEncoder:
    def __init__(self, cfg: Optional[PerceptionEncoderConfig] = None, tower: Optional[Callable] = None, device=None,
                 precision: str = "bf16"):
        self.cfg = cfg or PerceptionEncoderConfig()
        self.batch_size, self.dim = self.cfg.batch_size, self.cfg.dim
        self.normalize_feature, self.image_size = self.cfg.normalize_feature, self.cfg.image_size
        mode = self.cfg.interpolation_mode.upper()
        if mode not in _MODES:  # reference vision_encoder.py:93-99
            raise ValueError(f"Unsupported interpolation_mode: {self.cfg.interpolation_mode}")
        self.mode = _MODES[mode]
        self.device = device
        if tower is None:
            from .config import PE_VISION_CONFIGS
            if self.cfg.name in PE_VISION_CONFIGS:   # reference vision_encoder.py:86: pe.CLIP.from_config(cfg.name)
                from .vision_tower import PEVisionTower
                pe = PE_VISION_CONFIGS[self.cfg.name]
                if pe.output_dim != self.dim or pe.image_size != self.image_size:
                    raise ValueError(f"vision_encoder dim / image_size ({self.dim}, {self.image_size}) do not match the "
                                     f"named PE config {self.cfg.name!r} ({pe.output_dim}, {pe.image_size})")
                tower = PEVisionTower(name=self.cfg.name, precision=precision, device=device)
        self.tower = tower

    def load_state_dict(self, state_dict, strict: bool = True):
        """The `vision_encoder.*` tensors of a SAMAudio checkpoint (`model.visual.*` = the PE-Core tower)."""
        if not hasattr(self.tower, "load_state_dict"):
            raise NotImplementedError("the attached tower takes no weights")
        if self.device is not None and getattr(self.tower, "device", None) is None:
            self.tower.to(self.device)
        return self.tower.load_state_dict(state_dict, strict=strict)

    def transform(self, video: torch.Tensor) -> torch.Tensor:
        """uint8 / float frames [T, 3, H, W] -> normalised float [T, 3, S, S] (reference vision_encoder.py:91-113:
        Resize((S, S), interp) on a tensor -> x / 255 -> Normalize(0.5, 0.5)).  torchvision's tensor Resize
        antialiases bilinear / bicubic down-scaling by default; F.interpolate(antialias=True) is the same kernel."""
        x = video.float()
        if x.shape[-2:] != (self.image_size, self.image_size):
            kw = {"antialias": True, "align_corners": False} if self.mode != "nearest" else {}
            x = torch.nn.functional.interpolate(x, size=(self.image_size, self.image_size), mode=self.mode, **kw)
            if not video.is_floating_point():
                # torchvision's tensor Resize interpolates integer frames in float and casts back (round; uint8 is
                # clamped to 0..255, which also removes bicubic overshoot) before the `/ 255`
                x = x.round()
                if video.dtype == torch.uint8:
                    x = x.clamp(0, 255)
        return (x / 255.0 - 0.5) / 0.5

    def encode(self, frames: torch.Tensor) -> torch.Tensor:
        if self.tower is None:
            raise NotImplementedError(
                f"no vision tower for config name {self.cfg.name!r}: known PE configs are built on the HIP library "
                "(sam_audio_amd.vision_tower), anything else needs `tower=` (a callable encode_image(frames, "
                "normalize=...) -> [N, dim])")
        return self.tower(frames, normalize=self.normalize_feature)

    @torch.no_grad()
    def forward(self, videos: List[torch.Tensor]) -> torch.Tensor:
        """list of [T_i, 3, H, W] -> [B, max T_i, dim], zero-padded along time (reference vision_encoder.py:47-70)."""
        result = []
        for video in videos:
            video = self.transform(video.to(self.device) if self.device is not None else video)
            if self.batch_size > 0 and video.size(0) > self.batch_size:
                parts = [self.encode(video[i: i + self.batch_size]) for i in range(0, video.size(0), self.batch_size)]
                result.append(torch.cat(parts, dim=0))
            else:
                result.append(self.encode(video))
        return torch.nn.utils.rnn.pad_sequence(result, batch_first=True, padding_value=0.0)

    __call__ = forward
