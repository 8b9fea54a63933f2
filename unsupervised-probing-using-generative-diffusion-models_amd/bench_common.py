"""Rank plumbing shared by the secondary bench configs (c3, c5): one process per GPU, weak scaling (every rank owns windows of its own trajectory),
one all-gather of the per-window scalars per step, barrier + synchronize around the timed region, MAX over ranks."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def cpu_share() -> int:
    """Usable host cores: the cgroup CPU quota when there is one, else the affinity mask."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, int(os.environ.get("MPV_CPU_THREADS", cores))))


def init_group(backend, rank, world, device):
    """One process group per worker; the rendezvous and every collective give up after MPV_BENCH_DIST_TIMEOUT seconds."""
    import datetime
    timeout = datetime.timedelta(seconds=float(os.environ.get("MPV_BENCH_DIST_TIMEOUT", "600")))
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=timeout)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)


def measured_traffic(key, repo_root=None):
    """Per-kernel HBM traffic of one bench step from the committed PMC passes (profiles/*/pmc_traffic_<key>.json, written by
    tools/pmc_traffic.py: separate FETCH_SIZE and WRITE_SIZE passes, FETCH doubled as the guide prescribes for gfx950); newest round
    first, None when there is none."""
    import glob
    import json
    root = repo_root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "*", "pmc_traffic_%s.json" % key)), reverse=True):
        try:
            with open(f) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        rec["source"] = os.path.relpath(f, root)
        return rec
    return None


HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E, 8 TB/s peak (about 6.3 TB/s achievable)


def multi_kernel_roofline(tr, windows_per_s, achieved_tflops, peak_tflops, peak_name, kernels, extra=None):
    """Roofline object of a call made of many kernels (c3, c5).  With a committed PMC pass (`tr`, profiles/*/pmc_traffic_<cfg>.json) the
    bound is stated against HBM: achieved = measured HBM bytes per window x windows/s (every kernel of these schedules streams its operands
    once, so measured and algorithmic bytes coincide to within the L2 / Infinity-Cache hits the counters still see), `traffic` = those bytes
    per window; the matrix-pipe view of the same run is kept under `compute`.  Without a PMC pass only the compute view exists."""
    compute = {"achieved_tflops": achieved_tflops, "peak_tflops": peak_tflops, "peak": peak_name, "frac": achieved_tflops / peak_tflops}
    if tr and tr.get("bytes_per_window"):
        gbps = tr["bytes_per_window"] * windows_per_s / 1e9
        out = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": tr["bytes_per_window"],
               "kernel": kernels, "dominant_kernel": tr.get("dominant_kernel"), "compute": compute}
    else:
        out = {"bound": "mfma", "achieved": achieved_tflops, "peak": peak_tflops, "unit": "TFLOP/s", "frac": achieved_tflops / peak_tflops, "traffic": None,
               "kernel": kernels, "compute": compute}
    out.update(extra or {})
    return out


class Ranks:
    @classmethod
    def single(cls, device, backend="nccl"):
        """In-process secondary record of the default bench run: one rank, no process group."""
        self = cls.__new__(cls)
        self.world, self.rank, self.backend, self.device, self.own_group = 1, 0, backend, device, False
        return self

    def __init__(self, args):
        self.own_group = True
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
            init_group(self.backend, self.rank, self.world, self.device)

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
        if self.world > 1 and self.own_group:
            dist.destroy_process_group()
