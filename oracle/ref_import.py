"""Import the *reference's own* Python modules (read-only, from /root/reference) in this container.

TEST INFRASTRUCTURE ONLY - used by oracle/gen_golden.py to pin oracle/samaudio_oracle.py and to mint
tests/golden/*.npz.  /root/reference does not exist on the GPU box, so nothing at test / bench run
time imports this file.

The reference's third-party dependencies that are absent here (torchaudio, torchcodec, torchdiffeq,
dacvae, perception_models `core`, torchvision) are replaced by inert stub modules: they are only
needed for the import statements to succeed; none of their code is on the arithmetic we pin
(DiT, patcher, RoPE, AlignModalities, SAMAudio.forward / align_inputs glue, Batch.process_anchors).
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


class _Anything:
    """Class used for every attribute of a stub module (works as base class, type hint, callable)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed third-party dependency was called")

    @classmethod
    def from_config(cls, *a, **k):
        raise RuntimeError("stubbed third-party dependency was called")


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        obj = type(name, (_Anything,), {})
        setattr(self, name, obj)
        return obj


_STUBS = [
    "torchaudio", "torchaudio.functional", "torchcodec", "torchcodec.decoders", "torchcodec.encoders",
    "torchdiffeq", "dacvae", "torchvision", "torchvision.transforms", "torchvision.transforms.v2",
    "core", "core.audio_visual_encoder", "core.audio_visual_encoder.config",
    "core.audio_visual_encoder.transformer", "core.vision_encoder", "core.vision_encoder.pe",
]


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sam_audio"))


def import_reference():
    """Returns the imported reference package `sam_audio` (with stubbed third parties)."""
    if not available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not present (expected on the GPU box)")
    import transformers  # noqa: F401  (must probe torchvision & co. BEFORE the stubs exist)
    from transformers import AutoModel, AutoTokenizer, BatchFeature, ModernBertConfig  # noqa: F401
    for name in _STUBS:
        if name not in sys.modules:
            mod = _StubModule(name)
            mod.__path__ = []  # behave like a package so sub-imports resolve through sys.modules
            sys.modules[name] = mod
    for name in _STUBS:  # wire children as attributes of their parents
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, sys.modules[name])
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("sam_audio")
