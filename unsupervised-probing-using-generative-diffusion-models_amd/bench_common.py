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


FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, 256 CUs x 4 SIMD x 64 flop/clk x 2.4 GHz
F16_MATRIX_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16 / bf16 MFMA
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E, 8 TB/s peak (about 6.3 TB/s achievable)


def repo_root() -> str:
    return os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def lib_sha256(path=None):
    """sha256 of the HIP library this process would load (MPV_LIB or the in-tree libmpv_hip.so); None when it does not exist.  PMC traffic
    files carry the hash of the library they were measured on (tools/pmc_traffic.py) and are only quoted for that very binary."""
    import hashlib
    path = path or os.environ.get("MPV_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmpv_hip.so")
    try:
        h = hashlib.sha256()
        with open(path, "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 20), b""):
                h.update(blk)
        return h.hexdigest()
    except OSError:
        return None


def csrc_sha256(root=None):
    """sha256 over the kernel sources the library is built from (csrc/*.hip, csrc/*.h, include/mpv_hip.h, build.py's flags), in sorted order: the
    identity of the BINARY'S SOURCE, unchanged by a rebuild.  None when the sources are not there."""
    import glob
    import hashlib
    pkg = os.path.dirname(os.path.abspath(__file__))
    files = sorted(glob.glob(os.path.join(pkg, "csrc", "*.hip")) + glob.glob(os.path.join(pkg, "csrc", "*.h")))
    files += [os.path.join(root or repo_root(), "include", "mpv_hip.h"), os.path.join(pkg, "build.py")]
    h = hashlib.sha256()
    try:
        for f in files:
            h.update(os.path.basename(f).encode())
            with open(f, "rb") as fh:
                h.update(fh.read())
    except OSError:
        return None
    return h.hexdigest()


def measured_traffic(key, root=None, lib_hash=None, src_hash=None):
    """Per-kernel HBM traffic of one bench step from the committed PMC passes (profiles/*/pmc_traffic_<key>.json, written by
    tools/pmc_traffic.py: separate FETCH_SIZE and WRITE_SIZE passes, FETCH doubled as the guide prescribes for gfx950); newest round
    first, None when there is none.  A file measured on ANOTHER binary than the one loaded now is returned with `stale = True` and the caller
    must not quote its bytes as this run's traffic: the file's `_lib_sha256` must equal this library's, or -- a rebuild of unchanged sources on
    another box -- its `_csrc_sha256` the hash of the sources in the tree; a file that carries neither is always stale."""
    import glob
    import json
    root = root or repo_root()
    lib_hash = lib_hash or lib_sha256()
    if src_hash is None and os.environ.get("MPV_LIB") is None:          # (an explicitly chosen library is identified by its own hash only)
        src_hash = csrc_sha256()
    for f in sorted(glob.glob(os.path.join(root, "profiles", "*", "pmc_traffic_%s.json" % key)), reverse=True):
        try:
            with open(f) as fh:
                rec = json.load(fh)
        except (OSError, ValueError):
            continue
        rec["source"] = os.path.relpath(f, root)
        same_lib = bool(rec.get("_lib_sha256")) and rec.get("_lib_sha256") == lib_hash
        same_src = bool(rec.get("_csrc_sha256")) and src_hash is not None and rec.get("_csrc_sha256") == src_hash
        rec["stale"] = not (same_lib or same_src)
        return rec
    return None


def split_peak(gemm_mode):
    """(peak TFLOP/s for ALGORITHMIC flops, its description) of contractions evaluated in `gemm_mode`: the exact-fp32 matrix pipe, or the f16
    matrix pipe executing 3 / 2 / 1 MFMA products per algorithmic product (two-term split of both operands / of one / plain f16)."""
    if gemm_mode == "f32":
        return FP32_MATRIX_PEAK_TFLOPS, "fp32 matrix pipe 157.3 TFLOP/s"
    products = {"f16x3": 3, "f16x2": 2, "f16": 1}[gemm_mode]
    return F16_MATRIX_PEAK_TFLOPS / products, "dense f16 MFMA peak 2500 TFLOP/s / %d executed products per algorithmic product" % products


def multi_kernel_roofline(tr, windows_per_s, achieved_tflops, gemm_mode, kernels, extra=None):
    """Roofline object of a call made of many kernels (c3, c5, nss).  With a PMC pass of THIS binary (`tr`, profiles/*/pmc_traffic_<cfg>.json whose
    library hash matches) the bound is stated against HBM: achieved = measured HBM bytes per window x windows/s, `traffic` = those bytes per window;
    a pass of another binary is `traffic: None, traffic_stale: True` and the record falls back to the compute view.  The matrix-pipe view of the
    same run is always kept under `compute`, against the pipe that executes the products (`split_peak`); the ratio to the fp32 matrix peak only
    as `vs_fp32_matrix_peak`."""
    peak, peak_name = split_peak(gemm_mode)
    compute = {"achieved_tflops": achieved_tflops, "peak_tflops": peak, "peak": peak_name, "frac": achieved_tflops / peak,
               "vs_fp32_matrix_peak": achieved_tflops / FP32_MATRIX_PEAK_TFLOPS}
    if tr and tr.get("bytes_per_window") and not tr.get("stale"):
        gbps = tr["bytes_per_window"] * windows_per_s / 1e9
        out = {"bound": "hbm", "achieved": gbps, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS, "traffic": tr["bytes_per_window"],
               "traffic_source": tr.get("source"), "kernel": kernels, "dominant_kernel": tr.get("dominant_kernel"), "compute": compute}
    else:
        out = {"bound": "mfma", "achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tflops / peak, "traffic": None,
               "kernel": kernels, "compute": compute}
        if tr:
            out.update(traffic_stale=True, traffic_stale_source=tr.get("source"))
    out.update(extra or {})
    return out


# ---- one short line on stdout, everything else on stderr + a records file ----------------------------------------------------------------
VERBOSE_KEYS = ("traffic_detail", "note", "per_window_s", "peak_source", "parity", "kernels", "fx_peak", "passes_per_window", "traffic_stale_source",
                "dominant_kernel", "flop_per_sequence_forward", "flop_per_sequence_backward", "flop_per_row_forward")
LINE_LIMIT = 4000


def records_path():
    """Where the secondary records of a bench run go (JSON lines): MPV_BENCH_RECORDS, else gpurun_out/bench_records.jsonl under the repo."""
    p = os.environ.get("MPV_BENCH_RECORDS")
    if p:
        return p
    d = os.path.join(repo_root(), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        return os.path.join(d, "bench_records.jsonl")
    except OSError:
        return None


def emit_record(name, rec, fresh=False):
    """A secondary record (other GEMM modes, other configs / families, trained checkpoints ...): one JSON line on STDERR, prefixed
    `[bench record]`, and appended to the records file -- never on stdout, which carries the headline line alone."""
    import json
    import sys
    line = json.dumps({"record": name, **rec})
    print("[bench record] " + line, file=sys.stderr, flush=True)
    path = records_path()
    if path:
        try:
            with open(path, "w" if fresh else "a") as fh:
                fh.write(line + "\n")
        except OSError:
            pass


def compact(rec, limit=LINE_LIMIT):
    """The record as the ONE stdout line carries it: verbose members (per-kernel traffic tables, notes, long provenance strings) dropped at
    every depth, strings clipped; raises when the result still exceeds `limit` characters, so an over-long line fails here and not in the
    driver's parser (a 20 KB line was cut to its 8 000-character tail and could not be parsed)."""
    import json

    def walk(x):
        if isinstance(x, dict):
            return {k: walk(v) for k, v in x.items() if k not in VERBOSE_KEYS}
        if isinstance(x, (list, tuple)):
            return [walk(v) for v in x]
        if isinstance(x, str) and len(x) > 400:
            return x[:397] + "..."
        if isinstance(x, float):
            return float("%.6g" % x)
        return x

    out = walk(rec)
    n = len(json.dumps(out))
    if n > limit:
        raise ValueError("bench line is %d characters (limit %d): move members into emit_record()" % (n, limit))
    return out


def digest(rec):
    """value / unit / roofline fraction of a secondary record: what the headline line keeps of it (`other`)."""
    if not isinstance(rec, dict) or "value" not in rec:
        return {"error": (rec or {}).get("error", "no value")[:120]} if isinstance(rec, dict) else {"error": "no record"}
    roof = rec.get("roofline") or {}
    out = {"value": rec["value"], "unit": rec.get("unit")}
    if roof.get("frac") is not None:
        out.update(bound=roof.get("bound"), frac=roof["frac"])
    elif rec.get("frac") is not None:
        out.update(frac=rec["frac"])
    return out


def print_line(rec):
    """THE stdout line of a bench run."""
    import json
    print(json.dumps(compact(rec)), flush=True)


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
