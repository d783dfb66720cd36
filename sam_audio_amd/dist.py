"""Batch sharding across the GPUs of one node: one process per GPU, RCCL over xGMI.

separate() is embarrassingly parallel over clips (every op is per-sample: GroupNorm(1) statistics,
attention masks, ODE state - SURVEY.md §8e), so the only collectives are
  * start-up: broadcast of the checkpoint tensors from rank 0 (bucketed, ~256 MiB per collective so
    each xGMI link streams large messages), mirroring "replica per rank" of reference eval/main.py:53-63;
  * end of run: a gather of a few floats (reference eval/main.py:25-27 gathers a JSON blob).
Steady state has no cross-GPU traffic.  backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

from .processor import Batch

BUCKET_BYTES = 256 << 20


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_items: int, rank: int, world: int) -> range:
    """Contiguous split; the first (n % world) ranks take one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard_batch(batch: Batch, rank: int, world: int) -> Batch:
    """The clips of `batch` owned by `rank` (all candidates of a clip stay on one GPU)."""
    rows = list(shard_range(len(batch.descriptions), rank, world))
    idx = torch.tensor(rows, dtype=torch.long)

    def pick(t):
        return None if t is None else t[idx.to(t.device)]

    sizes = pick(batch.sizes)
    frames = int(sizes.max()) if len(rows) else 0
    samples = frames * batch.hop_length
    out = Batch.__new__(Batch)
    out.audios = pick(batch.audios)[..., :samples].contiguous()
    out.sizes, out.wav_sizes = sizes, pick(batch.wav_sizes)
    out.descriptions = [batch.descriptions[i] for i in rows]
    out.audio_pad_mask = pick(batch.audio_pad_mask)[:, :frames]
    out.masked_video = None if batch.masked_video is None else [batch.masked_video[i] for i in rows]
    out.hop_length, out.audio_sampling_rate = batch.hop_length, batch.audio_sampling_rate
    out.text_features, out.text_mask = pick(batch.text_features), pick(batch.text_mask)
    # rows of tensors that were range-checked on the host keep that guarantee (Batch.anchor_vocab_validated)
    out._set_anchor_tensors(pick(batch.anchor_ids), pick(batch.anchor_alignment)[:, :frames], getattr(batch, "anchor_vocab_validated", 0))
    out.anchors = None if batch.anchors is None else [batch.anchors[i] for i in rows]
    host = getattr(batch, "sizes_host", None)
    out.sizes_host = [host[i] for i in rows] if host is not None else [int(v) for v in sizes.tolist()]
    return out


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], src: int = 0, device=None) -> Dict[str, torch.Tensor]:
    """Rank `src` holds the checkpoint; everyone returns an identical copy on `device`.
    Tensors are packed into flat fp32/other-dtype buckets so that few, large collectives are issued."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return sd
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape), v.dtype) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    entries = meta[0]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    out: Dict[str, torch.Tensor] = {}
    bucket: List[Tuple[str, Tuple[int, ...], torch.dtype]] = []
    bucket_bytes = 0

    def flush():
        nonlocal bucket, bucket_bytes
        if not bucket:
            return
        dtype = bucket[0][2]
        # every tensor starts at a multiple of 256 bytes inside the bucket: the HIP library borrows these pointers and
        # requires 16-byte alignment (views at arbitrary element offsets were rejected by samaudio_set_tensor - found by the
        # 2-rank run on hardware, tools/r2_call17.sh)
        esz = torch.empty((), dtype=dtype).element_size()
        pad = max(1, 256 // esz)
        offs, total = [], 0
        for _, shape, _ in bucket:
            offs.append(total)
            total += (int(torch.Size(shape).numel()) + pad - 1) // pad * pad
        flat = torch.zeros(total, dtype=dtype, device=device)
        if rank == src:
            for (k, shape, _), off in zip(bucket, offs):
                flat[off:off + int(torch.Size(shape).numel())] = sd[k].reshape(-1).to(device)
        dist.broadcast(flat, src=src)
        for (k, shape, _), off in zip(bucket, offs):
            out[k] = flat[off:off + int(torch.Size(shape).numel())].reshape(shape)
        bucket, bucket_bytes = [], 0

    for k, shape, dtype in entries:
        nbytes = int(torch.Size(shape).numel()) * torch.empty((), dtype=dtype).element_size()
        if bucket and (bucket[0][2] != dtype or bucket_bytes + nbytes > BUCKET_BYTES):
            flush()
        bucket.append((k, shape, dtype))
        bucket_bytes += nbytes
    flush()
    return out


def gather_floats(values: List[float], dst: int = 0) -> Optional[List[List[float]]]:
    """End-of-run gather of a few per-rank numbers (timings); returns the table on `dst`."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(values)]
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.tensor(values, dtype=torch.float64, device=dev)
    table = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(table, mine)
    return [t.tolist() for t in table] if dist.get_rank() == dst else None
