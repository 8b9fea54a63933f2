"""Window sharding across ranks (one process per GPU) and reassembly of the MPV curve.

The path partitions cleanly: windows are independent (reference run_evaluation_cache:331-337
carries no state between them), so every rank takes a contiguous block of windows, holds a
replica of the <0.5 M weights, and the only exchange is ONE all-gather of the per-window
scalars (MPV, pred_mean): a few hundred bytes, latency-bound -- on GPUs this is a single
RCCL all-gather over xGMI, in the CPU tests gloo.  No collective touches the data path.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def active() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def rank() -> int:
    return dist.get_rank() if active() else 0


def world_size() -> int:
    return dist.get_world_size() if active() else 1


def is_primary() -> bool:
    return rank() == 0


def block_bounds(n_items: int, r: int, world: int) -> Tuple[int, int]:
    """Contiguous block of rank r: the first n % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = r * base + min(r, extra)
    return lo, lo + base + (1 if r < extra else 0)


def shard_bounds(n_items: int) -> Tuple[int, int]:
    return block_bounds(n_items, rank(), world_size())


def gather_curve(n_items: int, lo: int, hi: int, *local: torch.Tensor):
    """Each rank passes its [hi-lo] slices; every rank gets the full [n_items] numpy curves."""
    if not active():
        return tuple(t.detach().cpu().numpy() for t in local)
    world = world_size()
    width = -(-n_items // world)                      # every rank's block padded to the widest
    dev = local[0].device if dist.get_backend() == "nccl" else torch.device("cpu")
    send = torch.zeros(len(local), width, dtype=torch.float32, device=dev)
    for i, t in enumerate(local):
        send[i, : hi - lo] = t.detach().to(dev, torch.float32)
    recv = torch.empty(world, len(local), width, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send) if dist.get_backend() == "nccl" else \
        dist.all_gather(list(recv.unbind(0)), send)
    recv = recv.cpu().numpy()
    out = []
    for i in range(len(local)):
        parts = []
        for r in range(world):
            a, b = block_bounds(n_items, r, world)
            parts.append(recv[r, i, : b - a])
        out.append(np.concatenate(parts) if parts else np.zeros(0, np.float32))
    return tuple(out)
