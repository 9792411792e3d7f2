"""Multi-GPU self-play: games are independent, so rank r plays its own shard
with its own engine and no data-path collective; once per self-play round the
finished games are collected on the rank that owns the replay buffer with ONE
gather (RCCL over xGMI on GPUs, gloo in the CPU tests).

This replaces the per-game `buffer.save_game.remote(game, ...)` Ray calls of the
reference (Training/Gamer.py:95).
"""
import numpy as np
import torch
import torch.distributed as td

# field -> (dtype, trailing shape given (T, A, C, H, W))
FIELDS = ("states", "visits", "actions", "lengths", "outcomes", "tree_size", "n_children", "bias")


def shard_seeds(base_seed, games_per_rank, rank):
    """Global game index = rank * games_per_rank + g; its stream seed is base + index."""
    return base_seed + rank * games_per_rank


def pack(payload):
    """dict of tensors -> one contiguous uint8 tensor (+ layout for unpack)."""
    parts, layout, off = [], [], 0
    for k in FIELDS:
        t = payload[k].contiguous()
        b = t.view(torch.uint8).reshape(-1)
        pad = (-b.numel()) % 16
        if pad:
            b = torch.cat([b, torch.zeros(pad, dtype=torch.uint8, device=b.device)])
        layout.append((k, t.dtype, tuple(t.shape), off, t.numel() * t.element_size()))
        off += b.numel()
        parts.append(b)
    return torch.cat(parts), layout


def unpack(buf, layout):
    out = {}
    for k, dtype, shape, off, nbytes in layout:
        out[k] = buf[off:off + nbytes].view(dtype).reshape(shape)
    return out


def gather_payload(payload, world, rank, dst=0):
    """One gather of every rank's packed payload to `dst`.  Returns, on dst, a
    dict of tensors with the ranks' games concatenated along dim 0 (rank-major),
    elsewhere None."""
    buf, layout = pack(payload)
    if world == 1:
        return payload
    bufs = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    td.gather(buf, bufs, dst=dst)
    if rank != dst:
        return None
    per_rank = [unpack(b, layout) for b in bufs]
    return {k: torch.cat([p[k] for p in per_rank], 0) for k in FIELDS}


class ReplayGather:
    """Per-round collection of an engine's finished games on rank 0."""

    def __init__(self, engine, world, rank, dst=0):
        self.engine, self.world, self.rank, self.dst = engine, world, rank, dst
        self.last = None

    def gather(self):
        self.last = gather_payload(self.engine.export_device(), self.world, self.rank, self.dst)
        return self.last
