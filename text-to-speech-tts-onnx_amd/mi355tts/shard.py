"""Multi-GPU plumbing: one process per GPU, utterances sharded, no data-path collective.

The reference has no distributed code at all: utterances are independent (its
graphs are batch-1), so the scale-out axis is the utterance list.  The only collective is a one-off
broadcast of the packed weight blob (RCCL over xGMI when the backend is "nccl"; gloo in CPU tests);
results can optionally be gathered to rank 0.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced partition: the first (n % world) ranks get one extra item."""
    if world <= 0 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard arguments")
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def bucket_by_length(lengths: Sequence[int], max_batch: int) -> List[List[int]]:
    """Group utterance indices with equal max_duration into engine batches (the engine batches
    utterances of equal N; the reference has no batching, so no padding semantics exist to copy)."""
    groups = {}
    for i, n in enumerate(lengths):
        groups.setdefault(int(n), []).append(i)
    out = []
    for n in sorted(groups):
        idx = groups[n]
        for j in range(0, len(idx), max_batch):
            out.append(idx[j:j + max_batch])
    return out


def broadcast_blob(blob, src: int = 0, device=None):
    """Broadcast a flat fp32 weight blob from `src` to every rank.  `blob` is a numpy array on `src`
    and an int (element count) elsewhere.  Returns a numpy array on every rank."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return np.asarray(blob, dtype=np.float32)
    rank = dist.get_rank()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    n = torch.tensor([blob.size if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        t = torch.from_numpy(np.ascontiguousarray(blob, dtype=np.float32)).to(dev)
    else:
        t = torch.empty(int(n.item()), dtype=torch.float32, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_waveforms(local: Sequence[np.ndarray], dst: int = 0):
    """Gather per-rank lists of int16 waveforms to `dst` (returns the concatenated list there, None elsewhere)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local)
    objs = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(list(local), objs, dst=dst)
    if dist.get_rank() != dst:
        return None
    return [w for part in objs for w in part]
