"""Target sharding across GPUs and the one collective of the path.

Targets are independent (reference main.cpp:130: one private graph per
target), so the path shards with no data-path collective: rank r takes a
contiguous range of targets, balanced by alignment bytes (cost is proportional
to columns).  The only exchange is the final variable-length gather of the
FASTA payload on rank 0 (replacement for the Writer thread, main.cpp:164-175):
an all_gather of byte counts, then one gather of padded byte tensors -- RCCL
when the process group is "nccl", gloo on CPU in the tests.
"""
from __future__ import annotations

import numpy as np


def shard_ranges(weights, world: int):
    """Contiguous [begin, end) target ranges, one per rank, with near-equal
    total weight (weights[t] = alignment bytes of target t).  Every target
    lands in exactly one range; ranges are in rank order."""
    w = np.asarray(weights, dtype=np.float64)
    n = int(w.size)
    if world <= 0:
        raise ValueError("world must be positive")
    csum = np.concatenate([[0.0], np.cumsum(w)])
    total = csum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(csum, target, side="left"))
        b = min(max(b, bounds[-1]), n)
        bounds.append(b)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def gather_fasta(payload: bytes, dist, torch, local_rank=None, return_sizes=False):
    """Gather every rank's FASTA bytes on rank 0, in rank order (= global
    target order, since shards are contiguous).  Returns bytes on rank 0 and
    None elsewhere; with return_sizes also the byte count every rank announced."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    backend = dist.get_backend()
    dev = torch.device("cuda", local_rank if local_rank is not None else torch.cuda.current_device()) \
        if backend == "nccl" else torch.device("cpu")
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=dev)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    if rank == 0:
        parts = [torch.zeros(cap, dtype=torch.uint8, device=dev) for _ in range(world)]
        dist.gather(buf, parts, dst=0)
        out = b"".join(bytes(parts[r][:sizes[r]].cpu().numpy().tobytes()) for r in range(world))
        return (out, sizes) if return_sizes else out
    dist.gather(buf, None, dst=0)
    return (None, sizes) if return_sizes else None
