"""Rerankers for `separate(reranking_candidates > 1)` - the reference's `sam_audio.ranking` surface
(reference sam_audio/ranking/ranker.py:9-36, ranking/judge.py:11-42, ranking/__init__.py:15-30) for the rows this build
covers: the Judge and ensembles of rankers.  CLAP / ImageBind rankers wrap third-party
models that are not part of this build's environment: their configs are reported, not silently ignored.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch

from .config import JudgeRankerConfig


class Ranker:
    """forward(**kwargs) -> scores [batch, candidates] (higher = better); reference ranking/ranker.py:9-20."""

    def forward(self, **kwargs) -> torch.Tensor:
        raise NotImplementedError

    def __call__(self, **kwargs) -> torch.Tensor:
        return self.forward(**kwargs)


class EnsembleRanker(Ranker):
    """Weighted sum of rankers (reference ranking/ranker.py:23-36)."""

    def __init__(self, rankers: Sequence[Ranker], weights: Sequence[float]):
        assert len(rankers) == len(weights)
        self.rankers, self.weights = list(rankers), list(weights)

    def forward(self, **kwargs) -> torch.Tensor:
        result = None
        for weight, ranker in zip(self.weights, self.rankers):
            score = weight * ranker(**kwargs)
            result = score if result is None else result + score
        return result


class JudgeRanker(Ranker):
    """reference ranking/judge.py:11-42.  `model` is a loaded `SAMAudioJudgeModel`, `processor` a
    `SAMAudioJudgeProcessor`; from a config they are loaded from the local checkpoint directory.

    The reference expands the mixture to one copy per candidate before the processor (ranking/judge.py:31-33) and the
    Judge encodes every copy; here the mixture is handed over once per clip (`score_candidates`), same scores."""

    def __init__(self, config: Optional[JudgeRankerConfig] = None, model=None, processor=None, **model_kwargs):
        self.config = config
        if model is None or processor is None:
            from .judge import SAMAudioJudgeModel
            from .processor import SAMAudioJudgeProcessor
            path = config.checkpoint_or_model_id
            model = model or SAMAudioJudgeModel.from_pretrained(path, **model_kwargs)
            processor = processor or SAMAudioJudgeProcessor.from_pretrained(path)
        self.model, self.processor = model, processor

    @torch.inference_mode()
    def forward(self, input_audio: List[torch.Tensor], extracted_audio: List[torch.Tensor], descriptions: List[str],
                sample_rate: int = 48_000, **kwargs) -> torch.Tensor:
        bsz, ncandidates = len(extracted_audio), len(extracted_audio[0])
        # input_audio[b] is the mixture expanded to [candidates, n] (reference model.py:319-322): row 0 is the clip
        mixtures = [x[0][None] for x in input_audio]
        extracted = [x[None] for candidates in extracted_audio for x in candidates]
        processed = self.processor(text=list(descriptions), input_audio=mixtures, separated_audio=extracted,
                                   sampling_rate=sample_rate)
        return self.model.score_candidates(
            input_ids=processed["input_ids"], attention_mask=processed.get("attention_mask"),
            input_values=processed["input_values"], separated_values=processed["separated_values"],
            candidates=ncandidates, padding_mask=processed["padding_mask"]).view(bsz, ncandidates)


def create_ranker(config, **kwargs) -> Optional[Ranker]:
    """reference ranking/__init__.py:15-30."""
    if config is None:
        return None
    if isinstance(config, Ranker) or callable(config):
        return config
    if isinstance(config, JudgeRankerConfig):
        return JudgeRanker(config, **kwargs)
    kind = config.get("kind") if isinstance(config, dict) else getattr(config, "kind", None)
    if kind == "ensemble":
        pairs = config["rankers"].values() if isinstance(config, dict) else config.rankers.values()
        from .config import parse_ranker_config
        cfgs, weights = zip(*[(parse_ranker_config(c) if isinstance(c, dict) else c, w) for c, w in pairs])
        return EnsembleRanker([create_ranker(c, **kwargs) for c in cfgs], list(weights))
    raise NotImplementedError(
        f"ranker kind {kind!r} wraps a third-party model (CLAP / ImageBind / ...) that this build does not ship; "
        "attach a callable ranker to model.text_ranker / model.visual_ranker instead")
