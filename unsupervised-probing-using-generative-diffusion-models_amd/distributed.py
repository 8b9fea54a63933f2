"""Window sharding across ranks (one process per GPU) and reassembly of the MPV curve.

The path partitions cleanly: windows are independent (reference run_evaluation_cache:331-337
carries no state between them), so every rank takes a contiguous block of windows, holds a
replica of the <0.5 M weights, and the only exchange is ONE all-gather of the per-window
scalars (MPV, pred_mean): a few hundred bytes, latency-bound -- on GPUs this is a single
RCCL all-gather over xGMI, in the CPU tests gloo.  No collective touches the data path.

When the caller also wants the sample tensors themselves (the reference's `pred_future_list` and its list-of-tensors
cache hold one entry per window, diffusion_model_uncertainy.py:323-339) `gather_windows` reassembles them with a
second all-gather of the padded blocks, so every rank returns the same complete list and rank 0 writes the cache.

Second sharding level, for jobs with fewer windows than ranks: `sample_split()` decides to split the
SAMPLE axis instead; each rank draws S/world chains per (node, horizon, feature) row and `merge_row_moments` combines
the per-row (mean, M2) pairs with Chan's parallel update after one all-gather.
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import torch
import torch.distributed as dist


def active() -> bool:
    """A process group with more than one rank -- or, with MPV_FORCE_DIST=1, any initialised group: a world of ONE then takes the very
    collective branches (RCCL on device tensors under `nccl`) that N ranks take, which is how a one-GPU box exercises them
    (tests/test_rccl_world1_gpu.py)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("MPV_FORCE_DIST") == "1"


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
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
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


def gather_windows(n_items: int, lo: int, hi: int, local: torch.Tensor) -> torch.Tensor:
    """local [hi-lo, ...] (this rank's windows, any device) -> [n_items, ...] on the CPU, identical on every rank."""
    if not active():
        return local.detach().cpu()
    world = world_size()
    width = -(-n_items // world)
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    send = torch.zeros((width,) + tuple(local.shape[1:]), dtype=local.dtype, device=dev)
    send[: hi - lo] = local.detach().to(dev)
    recv = torch.empty((world,) + tuple(send.shape), dtype=local.dtype, device=dev)
    if nccl:
        dist.all_gather_into_tensor(recv, send)
    else:
        dist.all_gather(list(recv.unbind(0)), send)
    recv = recv.cpu()
    parts = []
    for r in range(world):
        a, b = block_bounds(n_items, r, world)
        parts.append(recv[r, : b - a])
    return torch.cat(parts, dim=0)


def barrier():
    if active():
        dist.barrier()


def sample_split(n_windows: int, n_samples: int, chunk: int, whole_chunks_only: bool = False) -> bool:
    """True when the sample axis, not the window axis, should be sharded: fewer windows than ranks, and each rank's
    n_samples / world samples are either whole `parallel_sample` chunks or (pointwise samplers only: not `whole_chunks_only`) a whole
    fraction of one chunk.  The chunk of DiffusionTS (one batch with shared loss normalisers) and of DiffSTG (one sequential pass of
    parallel graph copies) is not cut."""
    world = world_size()
    if not (active() and 0 < n_windows < world and n_samples % world == 0):
        return False
    per, chunk = n_samples // world, max(1, min(chunk, n_samples))
    return per % chunk == 0 or (not whole_chunks_only and chunk % per == 0)


def sample_bounds(n_samples: int) -> Tuple[int, int]:
    per = n_samples // world_size()
    return rank() * per, (rank() + 1) * per


def broadcast_int(value: int) -> int:
    """Rank 0's integer on every rank (a shared Philox seed)."""
    if not active():
        return int(value)
    nccl = dist.get_backend() == "nccl"
    t = torch.tensor([int(value)], dtype=torch.int64, device=torch.device("cuda", torch.cuda.current_device()) if nccl else "cpu")
    dist.broadcast(t, src=0)
    return int(t.item())


def gather_samples(local: torch.Tensor) -> torch.Tensor:
    """Each rank's [..., S/world] block of the sample axis -> the full [..., S] tensor on the CPU, rank order."""
    if not active():
        return local.detach().cpu()
    world = world_size()
    nccl = dist.get_backend() == "nccl"
    send = local.detach().contiguous()
    send = send.to(torch.device("cuda", torch.cuda.current_device())) if nccl else send.cpu()
    recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device)
    if nccl:
        dist.all_gather_into_tensor(recv, send)
    else:
        dist.all_gather(list(recv.unbind(0)), send)
    return torch.cat(list(recv.unbind(0)), dim=-1).cpu()


def merge_row_moments(mean: torch.Tensor, m2: torch.Tensor, n_local: int):
    """Per-row sample mean and centred second moment M2 = sum (y - mean)^2 of THIS rank's n_local samples
    (any shape, float64) -> population variance and mean over all ranks' samples, Chan et al. pairwise update applied
    in rank order so every rank gets bitwise the same result."""
    if not active():
        return m2 / n_local, mean
    world = world_size()
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    send = torch.stack([mean.double(), m2.double()]).to(dev).contiguous()
    recv = torch.empty((world,) + tuple(send.shape), dtype=torch.float64, device=dev)
    if nccl:
        dist.all_gather_into_tensor(recv, send)
    else:
        dist.all_gather(list(recv.unbind(0)), send)
    tot_mean, tot_m2, n = recv[0, 0], recv[0, 1], n_local
    for r in range(1, world):
        delta = recv[r, 0] - tot_mean
        tot_m2 = tot_m2 + recv[r, 1] + delta * delta * (n * n_local / (n + n_local))
        tot_mean = tot_mean + delta * (n_local / (n + n_local))
        n += n_local
    return tot_m2 / n, tot_mean
