"""Workspace / weight-registration helpers shared by the host classes that sit on the C ABI (judge.py, vision_tower.py,
t5_encoder.py, mbert_encoder.py)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict

import torch

from . import hip


def ensure_ws(owner, need: int, setter) -> None:
    ws = getattr(owner, "_workspace", None)
    if ws is None or ws.numel() < need + 256:
        fill = 255 if os.environ.get("SAMAUDIO_POISON") else None  # NaN bytes, see SAMAudio._ensure_workspace
        ws = (torch.full((need + 256,), fill, dtype=torch.uint8, device=owner.device) if fill is not None
              else torch.empty(need + 256, dtype=torch.uint8, device=owner.device))
        owner._workspace = ws
    base = ws.data_ptr()
    aligned = (base + 255) // 256 * 256
    hip.check(setter(C.c_void_p(aligned), ws.numel() - (aligned - base)))


def register(lib_set, handle, store: Dict[str, torch.Tensor], tensors: Dict[str, torch.Tensor]) -> None:
    for name, t in tensors.items():
        dt = hip.dtype_code(t.dtype)
        if t.data_ptr() % 16:   # a view into a larger buffer: the library needs 16-byte aligned pointers
            t = t.clone()
        store[name] = t  # keep alive: the library borrows the pointer
        hip.check(lib_set(handle, name.encode(), hip.ptr(t), dt, t.dim(), hip.shape_array(t.shape)))
