"""MI355X-native implementation of the SAMAudio.separate() hot path.

Public surface mirrors `sam_audio` (reference sam_audio/__init__.py:3-4) for that path:
SAMAudio, SAMAudioProcessor, Batch, SeparationResult.
"""
from .config import SAMAudioConfig, preset_config  # noqa: F401
from .processor import Batch, SAMAudioJudgeProcessor, SAMAudioProcessor  # noqa: F401


def __getattr__(name):
    # model.py loads libsamaudio_hip.so on construction; import it lazily so that host-only users
    # (processor, configs) work on machines without the built library.
    if name in ("SAMAudio", "SeparationResult", "DFLT_ODE_OPT"):
        from . import model
        return getattr(model, name)
    if name in ("SAMAudioJudgeModel", "SAMAudioJudgeOutput", "PEAudioFrame"):
        from . import judge
        return getattr(judge, name)
    raise AttributeError(name)


__all__ = ["SAMAudio", "SAMAudioProcessor", "SAMAudioJudgeProcessor", "SAMAudioJudgeModel", "Batch", "SeparationResult",
           "SAMAudioConfig", "preset_config"]
