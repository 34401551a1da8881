"""Utterance sharding across the GPUs of one node and the posterior gather.

The path shards by construction: windows, utterances and streams are independent units
(``utils/evaluate_models.py:70-88`` recomputes every window from scratch).  Each rank (one
process per GPU) processes its own utterances; the single exchange is a gather of the
posteriors (``torch.distributed`` - RCCL over xGMI with the ``nccl`` backend on the GPU box,
``gloo`` in the CPU tests).  Variable counts are handled by padding to the maximum and sending
the counts first.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def shard_by_length(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Length-sorted (longest first) round-robin deal: rank r gets utterances order[r::world]."""
    order = np.argsort(-np.asarray(lengths, dtype=np.int64), kind="stable")
    return [order[r::world_size].tolist() for r in range(world_size)]


def split_stream(n_posteriors: int, world_size: int) -> List[Tuple[int, int]]:
    """Contiguous posterior ranges of one long stream; posterior i needs mel rows
    [hop*i, hop*i + window) only, so ranks can work on overlapping row ranges independently."""
    base, rem = divmod(n_posteriors, world_size)
    out, lo = [], 0
    for r in range(world_size):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def gather_posteriors(local: np.ndarray, index: Sequence[int], n_total: int, group=None, device=None) -> np.ndarray:
    """All ranks contribute ``local[k]`` for global slot ``index[k]``; every rank receives the
    full ``[n_total]`` array.  One all_gather of counts + one all_gather of padded payloads."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    n = torch.tensor([len(index)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    mx = int(max(int(c.item()) for c in counts))
    payload = torch.zeros((mx, 2), dtype=torch.float64, device=dev)  # (slot, value)
    if len(index):
        payload[: len(index), 0] = torch.as_tensor(np.asarray(index, dtype=np.float64), device=dev)
        payload[: len(index), 1] = torch.as_tensor(np.asarray(local, dtype=np.float64), device=dev)
    bufs = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(bufs, payload, group=group)
    out = np.zeros(n_total, dtype=np.float32)
    for c, b in zip(counts, bufs):
        k = int(c.item())
        if k:
            b = b[:k].cpu().numpy()
            out[b[:, 0].astype(np.int64)] = b[:, 1].astype(np.float32)
    return out


def gather_values(local: np.ndarray, counts: Sequence[int], group=None, device=None) -> List[np.ndarray]:
    """Every rank contributes ``local`` (float32, ``counts[rank]`` values - every rank knows every rank's count, e.g. from a
    plan all ranks compute alike) and receives the list of all ranks' arrays: ONE all_gather of float32 payloads padded to
    the largest count, no index traffic (the sharded reference flow: each rank can work out which global slots any rank's
    values belong to)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    assert len(counts) == world and len(local) == counts[dist.get_rank(group)]
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    mx = int(max(counts)) if len(counts) else 0
    payload = torch.zeros(max(mx, 1), dtype=torch.float32, device=dev)
    if len(local):
        payload[: len(local)] = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float32), device=dev)
    bufs = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(bufs, payload, group=group)
    return [b[:int(c)].cpu().numpy() for b, c in zip(bufs, counts)]


def gather_values_t(local, counts: Sequence[int], group=None, device=None) -> list:
    """:func:`gather_values` on torch tensors: ``local`` (float32, on any device - normally the GPU that produced it) goes into
    ONE all_gather padded to the largest count and every rank's values come back as tensors on ``local``'s device (through
    ``device`` when the communicator lives elsewhere: gloo rehearsals exchange on the CPU).  With the ``nccl`` backend nothing
    touches the host: the values a rank's kernels produced are gathered, smoothed and swept where they are."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    assert len(counts) == world and local.numel() == counts[dist.get_rank(group)]
    dev = device if device is not None else ("cuda" if dist.get_backend(group) == "nccl" else "cpu")
    comm = torch.device(dev) if not isinstance(dev, torch.device) else dev
    if comm.type == "cuda" and comm.index is None and local.device.type == "cuda":
        comm = local.device
    mx = int(max(counts)) if len(counts) else 0
    payload = torch.zeros(max(mx, 1), dtype=torch.float32, device=comm)
    if local.numel():
        payload[: local.numel()] = local.to(comm)
    bufs = [torch.empty_like(payload) for _ in range(world)]
    dist.all_gather(bufs, payload, group=group)
    return [b[:int(c)].to(local.device) for b, c in zip(bufs, counts)]
