"""Data-parallel plumbing around the engine: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm;
"gloo" in the CPU tests).  The path shards by image: no collective inside the forward pass; the two
exchange steps the north_star names are the one-time weight broadcast and the per-batch gather of label maps.

The reference has no counterpart (single image, single GPU: src/process.cpp:70, src/main.cpp:148-164); its sequential
file loop is the place these helpers slot into.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int):
    """Contiguous split: rank r of R owns [lo, hi); the first n % R ranks own one extra item."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_blob(blob: bytes | None, nbytes: int, device, src: int = 0) -> bytes:
    """Rank `src` passes the weight-file bytes, everybody returns them (one contiguous broadcast)."""
    if dist.get_rank() == src:
        t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
        if t.numel() != nbytes:
            raise ValueError("blob length does not match nbytes")
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy().tobytes()


def gather_labels(local: torch.Tensor, counts, dst: int = 0):
    """Gather each rank's u8 label maps [n_r, H, W] to `dst` in rank order (ragged n_r allowed: padded to max)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    nmax = max(counts)
    pad = local
    if local.shape[0] < nmax:
        pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
