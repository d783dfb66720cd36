"""Host-side batching for SAMAudio.separate(): the reference's `SAMAudioProcessor` / `Batch` API
(reference sam_audio/processor.py:39-124, 158-260) for tensor inputs.

This is small integer / memcpy work and stays on the CPU.  File decoding in the reference goes
through torchaudio / torchcodec, which are not part of this build's environment: audio paths are accepted for
uncompressed PCM WAV at the model's sampling rate (standard library decoder), video paths raise.
Bit-exact parity of ``anchor_ids`` / ``anchor_alignment`` / ``sizes`` with the reference is pinned
by tests/golden/anchors.npz (minted from the reference's own Batch class).
"""
from __future__ import annotations

import json
import math
import os
from typing import List, Optional, Sequence, Tuple

import torch

from .config import SAMAudioConfig

Anchor = Tuple[str, float, float]
ANCHOR_VOCAB = {"<null>": 0, "+": 1, "-": 2, "<pad>": 3}


def mask_from_sizes(sizes: torch.Tensor) -> torch.Tensor:
    """[B] lengths -> [B, max] bool, True = valid (reference processor.py:127-128)."""
    steps = torch.arange(int(sizes.max()))
    return steps.unsqueeze(0) < sizes.unsqueeze(1)


def resample(wav: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6,
             rolloff: float = 0.99) -> torch.Tensor:
    """Band-limited sinc resampling of [..., samples], the algorithm `torchaudio.functional.resample` documents for its
    default method (reference processor.py:29-30 calls it for files whose rate differs from the model's 48 kHz):
    rates reduced by their gcd; for each of the `new` output phases a Hann-windowed sinc low-pass (cut-off = rolloff x the
    lower Nyquist rate, `lowpass_filter_width` zero crossings either side) sampled at the input positions; the filter bank is
    applied as one strided convolution (stride = reduced input rate) and the result is trimmed to ceil(new * n / orig)
    samples.  torchaudio is not part of this build's environment: restated from the published description, property-tested
    (tests/test_host_cpu.py), UNPINNED against torchaudio's own output.  Host-side preprocessing, runs once per file."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq <= 0 or new_freq <= 0:
        raise ValueError("sampling rates must be positive")
    if orig_freq == new_freq:
        return wav
    g = math.gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    taps = torch.arange(-width, width + o, dtype=torch.float64)[None, :] / o          # input sample positions (in seconds * g)
    phase = -torch.arange(n, dtype=torch.float64)[:, None] / n                       # output phase offsets
    t = ((phase + taps) * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / o)    # [n phases, 2*width + o]
    shape = wav.shape
    x = wav.reshape(-1, shape[-1])
    length = x.shape[-1]
    x = torch.nn.functional.pad(x, (width, width + o))
    y = torch.nn.functional.conv1d(x[:, None].to(torch.float64), kernel[:, None], stride=o)     # [rows, n, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)[:, : math.ceil(n * length / o)]
    return y.to(wav.dtype if wav.is_floating_point() else torch.float32).reshape(*shape[:-1], -1)


def load_wav(path: str, sampling_rate: int) -> torch.Tensor:
    """PCM WAV file -> float32 [channels, samples] in [-1, 1) with torchaudio.load's integer normalisation, resampled to
    the model's rate when the file's differs (reference processor.py:27-30: torchaudio.load + functional.resample).
    torchaudio / torchcodec are not part of this build's environment, so only what the standard library decodes is
    accepted - uncompressed PCM (8 / 16 / 24 / 32 bit)."""
    import wave
    import numpy as np
    try:
        with wave.open(path, "rb") as f:
            sr, ch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
            raw = f.readframes(n)
    except (wave.Error, EOFError) as exc:
        raise ValueError(f"{path}: only uncompressed PCM WAV files can be decoded without torchaudio ({exc})") from exc
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        x = (np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0).astype(np.float32)
    else:
        raise ValueError(f"{path}: unsupported sample width {width}")
    wav = torch.from_numpy(x.reshape(-1, ch).T.copy())
    return resample(wav, sr, sampling_rate) if sr != sampling_rate else wav


def batch_audio(audios: Sequence[torch.Tensor], audio_sampling_rate: int = 48_000) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mono mix-down + right zero-padding (reference processor.py:23-36)."""
    mono = []
    for a in audios:
        if isinstance(a, str):
            a = load_wav(a, audio_sampling_rate)
        if a.dim() != 2:
            raise ValueError(f"expected a (channels, samples) tensor, got shape {tuple(a.shape)}")
        mono.append(a.float().mean(0))
    sizes = torch.tensor([m.numel() for m in mono])
    out = torch.zeros(len(mono), 1, int(sizes.max()))
    for i, m in enumerate(mono):
        out[i, 0, : m.numel()] = m
    return out, sizes


class Batch:
    """Same fields as the reference Batch (processor.py:39-64) plus two optional ones,
    ``text_features`` / ``text_mask``, for callers that bring pre-computed T5 features (the t5-base
    tokenizer cannot be fetched offline)."""

    def __init__(self, audios: torch.Tensor, sizes: torch.Tensor, wav_sizes: torch.Tensor,
                 descriptions: List[str], hop_length: int, audio_sampling_rate: int,
                 anchors: Optional[List[List[Anchor]]] = None, audio_pad_mask: Optional[torch.Tensor] = None,
                 masked_video: Optional[List[torch.Tensor]] = None,
                 text_features: Optional[torch.Tensor] = None, text_mask: Optional[torch.Tensor] = None):
        assert audios.size(0) == len(descriptions), "one description per audio"
        self.audios = audios
        self.sizes = sizes
        self.wav_sizes = wav_sizes
        self.descriptions = descriptions
        self.audio_pad_mask = audio_pad_mask
        self.masked_video = masked_video
        self.hop_length = hop_length
        self.audio_sampling_rate = audio_sampling_rate
        self.text_features = text_features
        self.text_mask = text_mask
        # host copy of the frame counts (one read at construction, where the processor's tensors still live on the CPU): what
        # SAMAudio.separate() slices its outputs with - no device -> host read inside the timed path
        self.sizes_host: List[int] = [int(v) for v in sizes.tolist()]
        self.process_anchors(anchors)

    def _frame_of(self, seconds: float) -> int:
        return math.ceil(seconds * self.audio_sampling_rate / self.hop_length)

    def process_anchors(self, anchors: Optional[List[List[Anchor]]]) -> None:
        """Anchors -> (anchor_ids [B, 2+n], anchor_alignment [B, T]); reference processor.py:78-124.
        Slot 0 is <null>, slot 1 is <pad>; frame t of a span [ceil(start*25), ceil(end*25)) points at
        that anchor's slot, padded frames point at slot 1, everything else at slot 0."""
        n = len(self.audios)
        frames = self.audio_pad_mask.size(-1)
        alignment = torch.zeros(n, frames, dtype=torch.long)
        alignment[~self.audio_pad_mask.cpu()] = 1
        if anchors is None:
            ids = torch.tensor([[ANCHOR_VOCAB["<null>"], ANCHOR_VOCAB["<pad>"]]] * n, dtype=torch.long)
        else:
            rows = []
            for i, spans in enumerate(anchors):
                row = [ANCHOR_VOCAB["<null>"], ANCHOR_VOCAB["<pad>"]]
                for token, start, end in spans:
                    alignment[i, self._frame_of(start): self._frame_of(end)] = len(row)
                    row.append(ANCHOR_VOCAB[token])
                rows.append(row)
            width = max(len(r) for r in rows)
            ids = torch.full((n, width), ANCHOR_VOCAB["<pad>"], dtype=torch.long)
            for i, r in enumerate(rows):
                ids[i, : len(r)] = torch.tensor(r)
        # Built here, on the host, from vocabulary tokens and slot numbers, and CHECKED here on the CPU tensors (real checks, not
        # asserts: they survive `python -O`): every id is in [0, len(ANCHOR_VOCAB)), every alignment entry in [0, ids.size(1)).
        # SAMAudio.separate() then skips its device-side range check - four blocking device -> host reads - provided the vocabulary
        # the ids were checked against is the model's (`anchor_vocab_validated` == cfg.num_anchors + 1; reference model.py:61 would
        # raise inside gather / nn.Embedding).  Rebinding either tensor by hand (the property setters) drops the guarantee; moving
        # them with Batch.to() keeps it.
        if int(alignment.max()) >= ids.size(1) or int(alignment.min()) < 0:
            raise IndexError(f"anchor_alignment values must be in [0, {ids.size(1)})")
        if int(ids.max()) >= len(ANCHOR_VOCAB) or int(ids.min()) < 0:
            raise IndexError(f"anchor ids must be in [0, {len(ANCHOR_VOCAB)})")
        self._set_anchor_tensors(ids.to(self.audios.device), alignment.to(self.audios.device), len(ANCHOR_VOCAB))
        self.anchors = anchors

    # anchor tensors: plain attributes for readers; assignment from outside drops the host-side range guarantee
    def _set_anchor_tensors(self, ids, alignment, vocab_validated: int) -> None:
        self._anchor_ids, self._anchor_alignment, self.anchor_vocab_validated = ids, alignment, int(vocab_validated)

    @property
    def anchor_ids(self):
        return self._anchor_ids

    @anchor_ids.setter
    def anchor_ids(self, value) -> None:
        self._anchor_ids, self.anchor_vocab_validated = value, 0

    @property
    def anchor_alignment(self):
        return self._anchor_alignment

    @anchor_alignment.setter
    def anchor_alignment(self, value) -> None:
        self._anchor_alignment, self.anchor_vocab_validated = value, 0

    @property
    def anchors_validated(self) -> bool:
        """the anchor tensors are the ones process_anchors built and range-checked on the host (possibly moved by to())"""
        return getattr(self, "anchor_vocab_validated", 0) > 0

    def to(self, device) -> "Batch":
        for name in ("audios", "sizes", "wav_sizes", "audio_pad_mask", "text_features", "text_mask"):
            value = getattr(self, name)
            if value is not None:
                setattr(self, name, value.to(device))
        # (the same values on another device: the host-side range guarantee travels with them)
        self._set_anchor_tensors(self._anchor_ids.to(device), self._anchor_alignment.to(device),
                                 getattr(self, "anchor_vocab_validated", 0))
        if self.masked_video is not None:
            self.masked_video = [v.to(device) for v in self.masked_video]
        return self


def sample_video_frames(sizes: torch.Tensor, videos: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """One frame per latent step, uniformly spread (reference processor.py:147-153, tensor branch)."""
    picked = []
    for size, video in zip(sizes, videos):
        if isinstance(video, str):
            raise ValueError("video file paths need torchcodec, which this build does not ship")
        assert video.size(1) == 3, f"expected NCHW video, found {video.size(1)} channels"
        idx = torch.linspace(0, video.size(0) - 1, int(size)).round().long()
        picked.append(video[idx])
    return picked


class SAMAudioProcessor:
    def __init__(self, audio_hop_length: int, audio_sampling_rate: int):
        self.audio_hop_length = audio_hop_length
        self.audio_sampling_rate = audio_sampling_rate

    @classmethod
    def from_config(cls, cfg: SAMAudioConfig) -> "SAMAudioProcessor":
        return cls(cfg.audio_codec.hop_length, cfg.audio_codec.sample_rate)

    @classmethod
    def from_pretrained(cls, model_name_or_path: str) -> "SAMAudioProcessor":
        """Local directory holding the reference's config.json (processor.py:165-185); hub ids need
        network access this build does not have."""
        path = os.path.join(model_name_or_path, "config.json")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: only local checkpoints are supported offline")
        with open(path) as fin:
            return cls.from_config(SAMAudioConfig(**json.load(fin)))

    def feature_to_wav_idx(self, feature_idx):
        return feature_idx * self.audio_hop_length

    def wav_to_feature_idx(self, wav_idx):
        if torch.is_tensor(wav_idx):
            return torch.ceil(wav_idx / self.audio_hop_length)
        return math.ceil(wav_idx / self.audio_hop_length)

    def mask_videos(self, videos: Sequence[torch.Tensor], masks: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """Zero the masked object out of each frame (reference processor.py:197-204, tensor branch)."""
        return [v * m.eq(0) for v, m in zip(videos, masks)]

    def __call__(self, descriptions: List[str], audios: Sequence[torch.Tensor],
                 anchors: Optional[List[List[Anchor]]] = None,
                 masked_videos: Optional[Sequence[torch.Tensor]] = None,
                 text_features: Optional[torch.Tensor] = None,
                 text_mask: Optional[torch.Tensor] = None) -> Batch:
        assert len(descriptions) == len(audios)
        assert anchors is None or len(descriptions) == len(anchors)
        assert masked_videos is None or len(descriptions) == len(masked_videos)
        wavs, wav_sizes = batch_audio(audios, self.audio_sampling_rate)
        sizes = self.wav_to_feature_idx(wav_sizes)
        pad_mask = mask_from_sizes(sizes)
        video = None if masked_videos is None else sample_video_frames(sizes, masked_videos)
        return Batch(audios=wavs, sizes=sizes, wav_sizes=wav_sizes, descriptions=list(descriptions),
                     hop_length=self.audio_hop_length, audio_sampling_rate=self.audio_sampling_rate,
                     anchors=anchors, audio_pad_mask=pad_mask, masked_video=video,
                     text_features=text_features, text_mask=text_mask)


class JudgeBatch(dict):
    """Dict of tensors with `.to(device)` (stands in for transformers.BatchFeature at reference processor.py:333,358)."""

    def to(self, device) -> "JudgeBatch":
        return JudgeBatch({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in self.items()})


class SAMAudioJudgeProcessor(SAMAudioProcessor):
    """Reference processor.py:263-379, tensor branch (file paths need torchcodec, which this build does not ship).
    `tokenizer` is any callable with the Hugging Face tokenizer call signature."""

    def __init__(self, audio_hop_length: int, audio_sampling_rate: int, tokenizer=None):
        super().__init__(audio_hop_length, audio_sampling_rate)
        self.tokenizer = tokenizer

    @classmethod
    def from_pretrained(cls, model_name_or_path: str) -> "SAMAudioJudgeProcessor":
        from .config import SAMAudioJudgeConfig
        path = os.path.join(model_name_or_path, "config.json")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path}: only local checkpoints are supported offline")
        with open(path) as fin:
            cfg = SAMAudioJudgeConfig(**json.load(fin))
        import transformers
        tokenizer = transformers.AutoTokenizer.from_pretrained(model_name_or_path, local_files_only=True)
        return cls(cfg.audio_codec.hop_length, cfg.audio_codec.sample_rate, tokenizer)

    def _reflect_pad(self, wav: torch.Tensor) -> torch.Tensor:  # processor.py:285-291
        if wav.ndim == 1:
            wav = wav.unsqueeze(0)
        rem = wav.size(-1) % self.audio_hop_length
        if rem == 0:
            return wav
        return torch.nn.functional.pad(wav, (0, self.audio_hop_length - rem), mode="reflect")

    def _process_audio(self, raw_audio, sampling_rate: Optional[int] = None) -> JudgeBatch:  # processor.py:297-335
        if isinstance(raw_audio, str) or (isinstance(raw_audio, (list, tuple)) and raw_audio and isinstance(raw_audio[0], str)):
            raise ValueError("audio file paths need torchcodec, which this build does not ship; pass tensors")
        if sampling_rate is not None and sampling_rate != self.audio_sampling_rate:
            raise ValueError(
                f"The model corresponding to this feature extractor was trained using a sampling rate of "
                f"{self.audio_sampling_rate}; got {sampling_rate}.")
        items = list(raw_audio) if isinstance(raw_audio, (list, tuple)) else list(self._reflect_pad(raw_audio)[:, None])
        items = [self._reflect_pad(x).T for x in items]          # (num_samples, channels)
        for example in items:
            if example.ndim > 2:
                raise ValueError(f"Expected input shape (channels, num_samples), but got shape ({example.shape})")
        lengths = torch.tensor([x.size(0) for x in items])
        input_values = torch.nn.utils.rnn.pad_sequence(items, batch_first=True).transpose(1, 2)
        padding_mask = torch.arange(int(lengths.max()))[None] < lengths[:, None]
        return JudgeBatch(input_values=input_values, padding_mask=padding_mask)

    def __call__(self, text=None, input_audio=None, separated_audio=None, sampling_rate: Optional[int] = None,
                 **kwargs) -> JudgeBatch:  # processor.py:337-363
        batch = JudgeBatch()
        if text is not None:
            if self.tokenizer is None:
                raise RuntimeError("SAMAudioJudgeProcessor has no tokenizer (none can be downloaded offline)")
            batch.update(self.tokenizer(text, return_tensors="pt", padding="longest", max_length=512, truncation=True))
        if input_audio is not None:
            batch.update(self._process_audio(input_audio, sampling_rate))
        if separated_audio is not None:
            batch["separated_values"] = self._process_audio(separated_audio, sampling_rate)["input_values"]
        return batch
