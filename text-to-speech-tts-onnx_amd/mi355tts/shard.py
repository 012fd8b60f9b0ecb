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


def broadcast_blob_device(blob_t, src: int = 0, force: bool = False):
    """Device-resident form of broadcast_blob: `blob_t` is a float32 CUDA tensor on every rank (filled on `src`, empty elsewhere);
    it is broadcast IN PLACE over the process group's backend — "nccl" = RCCL over xGMI, the tensor never visits the host — and
    handed back for F5Engine(blob_device=...) / BigVGANVocoder(blob_device=...), which build their engines straight from HBM.
    gloo (the one-GPU plumbing tests) stages through host memory, because its device-tensor support is not a given on ROCm.
    force: issue the collective on a one-rank group too (`bench.py --gpus 1 --force-collective`, tests/test_gpu_multirank.py):
    the RCCL library path — communicator, stream, kernel launch on the blob — runs on a one-GPU box as it does on eight."""
    import torch.distributed as dist
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return blob_t
    if dist.get_backend() == "nccl":
        dist.broadcast(blob_t, src=src)
        return blob_t
    h = blob_t.cpu()
    dist.broadcast(h, src=src)
    blob_t.copy_(h)
    return blob_t


def assert_one_device_per_rank(device_index: int, allow_shared: bool = False):
    """Every rank of a node must drive its own GPU: gathers (hostname, PCI bus id of this rank's device) and fails loudly on
    every rank if two ranks resolved to the same physical device (e.g. LOCAL_RANK not honoured, HIP_VISIBLE_DEVICES collapsing
    the ranks onto device 0) — a scaling line measured that way would be N ranks time-slicing one GPU.  Returns the id list."""
    import socket
    import torch.distributed as dist
    from . import _lib
    me = (socket.gethostname(), _lib.device_pci_bus_id(device_index))
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [me]
    ids = [None] * dist.get_world_size()
    dist.all_gather_object(ids, me)
    if len(set(ids)) != len(ids) and not allow_shared:
        raise RuntimeError(f"ranks share a physical GPU: {ids} (one process per GPU is the contract; "
                           "MI355TTS_BENCH_ONE_GPU=1 allows it for plumbing tests)")
    return ids


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
