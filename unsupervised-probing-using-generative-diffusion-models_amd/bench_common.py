"""Rank plumbing shared by the secondary bench configs (c3, c5): one process per GPU, weak scaling (every rank owns windows of its own trajectory),
one all-gather of the per-window scalars per step, barrier + synchronize around the timed region, MAX over ranks."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


class Ranks:
    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a ROCm device: the path has no CPU implementation")
        self.backend = args.backend
        if self.backend == "gloo":                                 # rehearsal: every rank on the one visible device
            local %= torch.cuda.device_count()
        torch.cuda.set_device(local)
        self.device = torch.device("cuda", local)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29535")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)

    def fence(self):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize(self.device)

    def gather(self, curve: torch.Tensor) -> torch.Tensor:
        """[2, W] on device -> [world, 2, W] on the host (the reassembled curves; a host copy is the step's sync point)."""
        if self.world == 1:
            return curve.unsqueeze(0).cpu()
        if self.backend == "nccl":
            out = torch.empty((self.world,) + tuple(curve.shape), device=self.device, dtype=curve.dtype)
            dist.all_gather_into_tensor(out, curve.contiguous())
            return out.cpu()
        parts = [torch.empty(tuple(curve.shape), dtype=curve.dtype) for _ in range(self.world)]
        dist.all_gather(parts, curve.cpu())
        return torch.stack(parts)

    def max_time(self, elapsed: float) -> float:
        if self.world == 1:
            return elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.world > 1:
            dist.destroy_process_group()
