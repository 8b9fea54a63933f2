#!/usr/bin/env python3
"""Benchmark of the rolling-window MPV sampling path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic trajectory: for every rolling window
(181 of them at BASELINE.json configs[1]) scaler -> sigma head -> N=32 chains x T=50 reverse
steps -> inverse scaling + variance reduction -> MPV scalar on the host.  Inputs (sub-sampled
series, weights) are resident in HBM before the timed region.  With N ranks every rank owns a
trajectory of the same size (weak scaling: windows shard with no data-path collective) and one
all-gather reassembles the curves.

Prints ONE short JSON line on stdout (rank 0; < 4 KB, checked before it is printed): metric/value/unit as BASELINE.json names them, plus
  roofline     -- the chain kernel against the matrix-pipe peak of the mode it runs in, timed live with HIP events
  cpu_baseline -- the CPU oracle (a torch-CPU port of the reference path) on a bounded sample
  mpv_*_err    -- MPV of the HIP path vs that CPU run on the SAME noise
  other        -- value / unit / roofline fraction of every secondary record
The secondary records themselves (other GEMM modes, sample-writing path, trained checkpoints, full NsDiff with f(x), BASELINE
configs[2..4], the NsDiff_spatial and TMDM families; each with its own roofline and CPU baseline) go to STDERR as `[bench record] {...}`
lines and to gpurun_out/bench_records.jsonl (MPV_BENCH_RECORDS overrides) -- never into the stdout line.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

def log(msg):
    print("[bench %7.1fs] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def cpu_share() -> int:
    from mpv_amd.bench_common import cpu_share as share
    return share()


FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs x 4 SIMD x 64 flop/clk x 2.4 GHz
F16_MATRIX_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA

CFG = dict(nodes=100, ticks=100_000, sampling_t=10, windows=100, pred_len=100, rolling=50, window_step=5,
           steps=50, n_z=32, parallel_sample=32, nf=1)


def build_model(device, series):
    from mpv_amd.models.models import diffusion_models
    from mpv_amd import synthetic
    net_param = dict(device=device, scaler_type="StandardScaler", task_model="NsDiff_model_variants",
                     rolling_length=CFG["rolling"], diffusion_schedule="linear", diffusion_steps=CFG["steps"],
                     beta_start=1e-4, beta_end=0.02, n_z_samples=CFG["n_z"], parallel_sample=CFG["parallel_sample"],
                     windows=CFG["windows"], pred_len=CFG["pred_len"], dataset_nf=CFG["nf"])
    model = diffusion_models("NsDiff_model_variants", net_param, train_model_select="cond_var").to(device).eval()
    return synthetic.seeded_nsdiff_state(model, series, seed=123)


def launch_workers(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh worker processes (one per GPU) BEFORE anything in this
    process touches the GPU, relay rank 0's JSON line, fail if any worker fails.  Nothing is exec'd.  All children are polled: the
    first non-zero exit (or the overall deadline, MPV_BENCH_LAUNCH_TIMEOUT seconds) terminates the rest, so a rank that dies before
    the rendezvous shows up as a failure instead of a hang; every rank's stderr is relayed."""
    import socket
    import subprocess
    import tempfile
    deadline = time.monotonic() + float(os.environ.get("MPV_BENCH_LAUNCH_TIMEOUT", "3000"))
    with socket.socket() as sk:                 # a free port now; the workers' rendezvous carries its own timeout should it be taken meanwhile
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs, outs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), MPV_BENCH_WORKER="1")
        outs.append(tempfile.TemporaryFile(mode="w+") if r == 0 else None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=outs[r] if r == 0 else subprocess.DEVNULL))          # stderr: inherited, every rank
    codes = [None] * len(procs)
    failed = None
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0) and failed is None:
                    failed = "rank %d exited with code %d" % (r, codes[r])
        if failed is None and time.monotonic() > deadline:
            failed = "workers still running at the launch deadline"
        if failed is not None:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.2)
    outs[0].seek(0)
    for line in outs[0].read().splitlines():            # rank 0 prints exactly one JSON record; drop library banners (gloo)
        if line.startswith("{"):
            print(line, flush=True)
        elif line.strip():
            print(line, file=sys.stderr)
    outs[0].close()
    if failed is not None:
        print("bench workers failed: %s (exit codes %s)" % (failed, codes), file=sys.stderr)
        return 1
    return 0


def init_group(backend, rank, world, device):
    from mpv_amd.bench_common import init_group as init
    init(backend, rank, world, device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=("c2", "c3", "c4", "c5", "nss", "tmdm"), default="c2",
                    help="c2 (default, the headline): NsDiff 100-node biomass, N=32, T=50; c3: DiffusionTS 100-node SIS, N=32, T=100; "
                         "c4: NsDiff 500-node Wilson-Cowan, N=64, T=50, ONE trajectory split over the ranks (strong scaling); "
                         "c5: DiffSTG 200-node small-world biomass, N=32, 50 inference steps")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=4, help="windows of the workload timed on the CPU oracle (each ~25 s on 16 threads)")
    ap.add_argument("--gemm", choices=("f16x3", "f32", "f16x2", "f16"), default="f16x3",
                    help="evaluation of the two 128x128 layers for the HEADLINE number: split-f16 matrix cores, fp32-equivalent "
                         "(default), exact fp32 MFMA, or the reduced-precision operand modes f16x2 / f16 (fp32 accumulate)")
    ap.add_argument("--host-data", action="store_true",
                    help="generate the synthetic trajectory with the numpy Euler-Maruyama on the host instead of the HIP kernel")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the extra, separately reported runs (other GEMM modes, sample-writing path, trained checkpoints, "
                         "the other BASELINE configs; N=1 only)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` records (BASELINE configs[2..4] = c3, c4, c5 at full size, each with its roofline and CPU baseline)")
    ap.add_argument("--store-samples", action="store_true",
                    help="headline step writes the [W,V,O,F,S] sample tensor and reduces it in a second kernel (the round-1 path)")
    ap.add_argument("--max-windows", type=int, default=None, help="profiling aid: cut the trajectory to this many windows "
                    "(the reported metric is then NOT the BASELINE workload and says so)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl", help="gloo: rehearse the N>1 launch path on a 1-GPU box")
    ap.add_argument("--dts-windows", type=int, default=1, help="--config c3: rolling windows per step (each 3200 sequences)")
    ap.add_argument("--stg-windows", type=int, default=4, help="--config c5: rolling windows per step (each 6400 graph-node rows)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(launch_workers(args))
    if os.environ.get("MPV_BENCH_FAIL_RANK") is not None and os.environ.get("MPV_BENCH_FAIL_RANK") == os.environ.get("RANK"):
        raise SystemExit(3)                     # test hook (tests/test_bench_launch_gpu.py): a rank that dies before the rendezvous
    if args.config == "c3":
        from mpv_amd.bench_dts import main_c3
        return main_c3(args, log)
    if args.config == "c5":
        from mpv_amd.bench_stg import main_c5
        return main_c5(args, log)
    if args.config == "c4":
        from mpv_amd.bench_c4 import main_c4
        return main_c4(args, log)
    if args.config in ("nss", "tmdm"):             # the `other_families` records on their own (N = 1; profiling passes: tools/pmc_pass.sh nss)
        import mpv_amd  # noqa: F401
        from mpv_amd import bench_families as BF
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU")
        fn = BF.record_nsdiff_spatial if args.config == "nss" else BF.record_tmdm
        from mpv_amd.bench_common import emit_record, print_line
        rec = fn(torch.device("cuda", 0), log, cpu=not args.no_cpu_baseline)
        rec.update(n_gpus=1, steps=2, warmup=1, higher_is_better=True, scaling="weak", vs_baseline=None)
        emit_record(args.config, rec, fresh=True)
        print_line(rec)
        return 0

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the path has no CPU implementation")
    if args.backend == "gloo":                                     # rehearsal: every rank on the one visible device
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("MPV_BENCH_FORCE_DIST") == "1"     # force: smoke the RCCL path on one GPU
    if use_dist:
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":    # keeps RCCL's version banner off stdout: rank 0 prints
            del os.environ["NCCL_DEBUG"]                             # exactly one line there, the JSON record
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        init_group(args.backend, rank, world, device)

    from mpv_amd import synthetic
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H

    # ---- inputs, resident before timing -------------------------------------------------
    log("rank %d/%d on %s: generating trajectory" % (rank, world, device))
    stride = H.sampling_interval_from_t(CFG["sampling_t"])
    if args.host_data:
        record = synthetic.biomass_trajectory(CFG["nodes"], CFG["ticks"], seed=1000 + rank)
        full = H.normalize_time_series(record["ys_dynamic"], dynamic_type="biomass")       # [V, ticks, 1]
        sampled, _ = H.sample_time_series(full, record["ts_dynamic"], CFG["sampling_t"])    # [V, 1000, 1]
        sampled = sampled.contiguous()
    else:                                       # Euler-Maruyama on device, sub-sampling fused (csrc/biomass_sim.hip)
        from mpv_amd import _native
        t_gen = time.perf_counter()
        gen_ctx = _native.Context(device)
        series_gen, _ = synthetic.biomass_trajectory_device(gen_ctx, CFG["nodes"], CFG["ticks"], seed=1000 + rank, sampling_stride=stride)
        sampled = series_gen.cpu().contiguous()
        log("device trajectory: %d ticks x %d nodes in %.3f s" % (CFG["ticks"], CFG["nodes"], time.perf_counter() - t_gen))
        gen_ctx.close()
    n_windows = H.sliding_window_count(sampled.shape[1], CFG["windows"], CFG["window_step"])
    if args.max_windows is not None:
        n_windows = min(n_windows, args.max_windows)
    model = build_model(device, sampled)
    model.gemm_mode = args.gemm
    ctx = model.native_context()
    series_dev = sampled.to(device)
    mean_dev, std_dev = model.scaler_mean.float().contiguous(), model.scaler_std.float().contiguous()
    S, P = CFG["n_z"], CFG["parallel_sample"]

    def step(seed, gemm=args.gemm, store=args.store_samples):
        gx = ctx.sigma_head(series_dev, n_windows, 0, CFG["window_step"], mean_dev, std_dev, add_eps=model.EPS)
        if store:      # sample tensor to HBM (232 MB), then the stand-alone reduction re-reads it
            samples = ctx.sample(gx, None, S, P, seed=seed, gemm=gemm)
            mpv, pm, _ = ctx.reduce(samples, std_dev, mean_dev)
        else:          # summary fused into the sampler's epilogue: no sample tensor at all
            _, mpv, pm, _ = ctx.sample(gx, None, S, P, seed=seed, gemm=gemm, summary=True, want_samples=False, std=std_dev, mean=mean_dev)
        curve = torch.stack([mpv, pm])                                                      # [2, W]
        if use_dist:
            if args.backend == "nccl":
                out = torch.empty(world, 2, n_windows, device=device)
                dist.all_gather_into_tensor(out, curve)
                curve = out
            else:
                parts = [torch.empty(2, n_windows) for _ in range(world)]
                dist.all_gather(parts, curve.cpu())
                curve = torch.stack(parts)
        return curve.cpu()                                                                   # host copy = sync

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    def timed(n_steps, seed0, **kw):
        ctx.timing_enable(True)
        ctx.timing()                                                                         # drop earlier records
        fence()
        t0 = time.perf_counter()
        for i in range(n_steps):
            curve = step(seed0 + i, **kw)
        fence()
        dt = time.perf_counter() - t0
        kms, n = ctx.timing()
        ctx.timing_enable(False)
        return dt, kms, n, curve

    log("%d windows resident; warmup" % n_windows)
    for i in range(args.warmup):
        step(i)
        log("warmup step %d done" % i)
    elapsed, kernel_ms, launches, curve = timed(args.steps, 100)
    log("timed %d steps in %.3f s; chain kernel avg %.2f ms" % (args.steps, elapsed, kernel_ms))
    if use_dist:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert bool(torch.isfinite(curve).all()), "non-finite MPV curve"

    total_windows = world * n_windows * args.steps
    value = total_windows / elapsed
    row_steps = n_windows * CFG["nodes"] * CFG["pred_len"] * S * CFG["steps"]               # chain-steps per launch
    flop_per_launch = row_steps * 256 * (256 + 5 * CFG["nf"])
    achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    out = {
        "metric": "rolling windows/sec (N=32, T=50) + MPV max-abs err vs CPU ref",
        "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.gemm],
        "data": "synthetic",
        "config": {"workload": "NsDiff cond_var on 100-node BA-graph biomass trajectory (BASELINE configs[1]): "
                               "windows=100, pred_len=100, rolling=50, N=32 samples, T=50 steps, {} windows per "
                               "trajectory, one trajectory per GPU per step".format(n_windows),
                   "windows_per_step_per_gpu": n_windows, "truncated_for_profiling": args.max_windows is not None, "noise": "on-device Philox4x32-10",
                   "weights": "seeded default init (shipped checkpoints pin T=20)",
                   "summary": "sample tensor written + stand-alone reduction" if args.store_samples else
                              "MPV reduction fused into the sampler epilogue, no sample tensor",
                   "parallelism": "windows sharded over %d rank(s), one all-gather of the curve" % world},
        "roofline": roofline(args.gemm, achieved, kernel_ms, launches, flop_per_launch, n_windows, args.store_samples),
    }
    if use_dist:
        out["collective"] = {"backend": dist.get_backend(), "forced_world_of_one": world == 1}

    # ---- separately reported runs: the other GEMM modes, the sample-writing path, trained checkpoints, full NsDiff, the
    # other BASELINE configs and model families.  Each is ONE record on stderr + in the records file; the stdout line keeps a digest.
    from mpv_amd.bench_common import digest, emit_record, print_line, records_path
    if rank == 0:
        emit_record("run", {"argv": sys.argv[1:], "n_gpus": world, "records_file": records_path()}, fresh=True)
    extras = world == 1 and not args.no_other_modes
    others = [g for g in ("f32", "f16x2", "f16") if g != args.gemm] if extras else []
    other_modes, other = {}, {}
    for g in others:
        n_steps = max(1, min(args.steps, 3 if g != "f32" else 2))
        step(7, gemm=g)
        dt, kms, _, _ = timed(n_steps, 200, gemm=g)
        products = {"f16x2": 2, "f16": 1, "f32": 0}[g]
        peak = FP32_MATRIX_PEAK_TFLOPS if g == "f32" else F16_MATRIX_PEAK_TFLOPS / products
        ach = flop_per_launch / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        other_modes[g] = {"value": n_windows * n_steps / dt, "unit": "windows/s", "steps": n_steps, "ms_per_step": dt / n_steps * 1e3,
                          "dtype": DTYPES[g], "kernel_avg_ms": kms, "achieved_tflops": ach, "mfma_peak_tflops": peak, "frac": ach / peak}
        log("mode %s: %.1f windows/s" % (g, other_modes[g]["value"]))
    if extras:
        step(8, store=not args.store_samples)
        dt, kms, _, _ = timed(3, 300, store=not args.store_samples)
        rec = {"summary": "sample tensor written + stand-alone reduction" if not args.store_samples else "fused",
               "value": n_windows * 3 / dt, "unit": "windows/s", "kernel_avg_ms": kms}
        emit_record("other_summary_path", rec)
        other["other_summary_path"] = digest(rec)
        for key, fn in (("trained_checkpoints", lambda: trained_checkpoint_records(device)),
                        ("full_nsdiff", lambda: full_nsdiff_record(device, sampled, n_windows))):
            try:                                   # secondary records never take the headline line down with them
                rec = fn()
            except Exception as exc:               # noqa: BLE001
                rec = {"error": "%s: %s" % (type(exc).__name__, exc)}
                log("%s failed: %s" % (key, rec["error"]))
            emit_record(key, rec)
            other[key] = digest(rec) if key == "full_nsdiff" else {k: digest(v) for k, v in rec.items()} if "error" not in rec else digest(rec)

    if extras and not args.no_other_configs and args.max_windows is None:
        for key, rec in list(other_config_records(args, device).items()) + list(other_family_records(args, device).items()):
            emit_record(key, rec)
            other[key] = digest(rec)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        parity = cpu_baseline_and_parity(model, sampled, args.cpu_windows, [args.gemm] + others)
        for g, err in parity.pop("per_mode").items():
            if g == args.gemm:
                out.update(err)
            else:
                other_modes[g].update(err)
        out.update(parity)
    for g, rec in other_modes.items():
        emit_record("mode_" + g, rec)
        other["mode_" + g] = digest(rec)
    if other:
        out["other"] = other
        out["records_file"] = os.path.relpath(records_path(), str(ROOT)) if records_path() else None
    if rank == 0:
        emit_record("headline", out)               # the complete headline record (with traffic_detail, notes)
        print_line(out)                            # ... and its short form: THE stdout line
    if use_dist:
        dist.destroy_process_group()


def other_config_records(args, device):
    """Secondary records: the other BASELINE configs that fit one GPU, each at FULL size through its own bench module (the same code
    `--config c3|c4|c5` runs), with its roofline, PMC traffic (profiles/*/pmc_traffic_<cfg>.json) and bounded CPU baseline.  Cheapest
    first; a failure is recorded and never takes the headline line down."""
    import types
    from mpv_amd.bench_common import Ranks
    recs = {}
    plan = (("c5", "mpv_amd.bench_stg", "record_c5", dict(steps=3, warmup=1)),
            ("c4", "mpv_amd.bench_c4", "record_c4", dict(steps=1, warmup=1)),
            ("c3", "mpv_amd.bench_dts", "record_c3", dict(steps=1, warmup=1)))
    for key, mod, fn, kw in plan:
        sub = types.SimpleNamespace(gpus=1, backend=args.backend, gemm="f16x3", no_cpu_baseline=args.no_cpu_baseline, dts_windows=1,
                                    stg_windows=4, **kw)
        t0 = time.perf_counter()
        try:
            import importlib
            recs[key] = getattr(importlib.import_module(mod), fn)(sub, log, Ranks.single(device, args.backend))
            log("other config %s: %.4g %s (%.1f s in all)" % (key, recs[key]["value"], recs[key]["unit"], time.perf_counter() - t0))
        except Exception as exc:                   # noqa: BLE001
            recs[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            log("other config %s failed: %s" % (key, recs[key]["error"]))
        torch.cuda.empty_cache()
    return recs


def other_family_records(args, device):
    """Secondary records of the model families BASELINE.json's configs do not name (NsDiff_spatial, TMDM): mpv_amd/bench_families.py."""
    recs = {}
    for key, fn in (("nsdiff_spatial", "record_nsdiff_spatial"), ("tmdm", "record_tmdm")):
        try:
            import importlib
            recs[key] = getattr(importlib.import_module("mpv_amd.bench_families"), fn)(device, log, cpu=not args.no_cpu_baseline)
        except Exception as exc:                   # noqa: BLE001
            recs[key] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            log("other family %s failed: %s" % (key, recs[key]["error"]))
        torch.cuda.empty_cache()
    return recs


def full_nsdiff_record(device, sampled, n_windows):
    """Secondary record: the same workload with `task_model: NsDiff` / `NsDiff_model`, i.e. with the mean backbone f(x) in front of
    the sampler (the shipped NsDiff YAMLs' ns-Transformer: d_model 512, 8 heads, d_ff 256, 4 + 2 layers; seeded default init).  One
    launch sequence per trajectory: f(x) over all windows (csrc/nst.hip), g(x), fused sampler + summary."""
    from mpv_amd import synthetic
    from mpv_amd.models.models import diffusion_models
    net_param = dict(device=device, scaler_type="StandardScaler", task_model="NsDiff", rolling_length=CFG["rolling"], diffusion_schedule="linear",
                     diffusion_steps=CFG["steps"], beta_start=1e-4, beta_end=0.02, n_z_samples=CFG["n_z"], parallel_sample=CFG["parallel_sample"],
                     windows=CFG["windows"], pred_len=CFG["pred_len"], dataset_nf=CFG["nf"], load_pretrain=False, d_model=512, n_heads=8, d_ff=256,
                     e_layers=4, d_layers=2, activation="gelu", p_hidden_dims=[64, 64], p_hidden_layers=2, output_attention=False)
    model = diffusion_models("NsDiff", net_param, train_model_select="NsDiff_model").to(device).eval()
    synthetic.seeded_nsdiff_state(model, sampled, seed=321)
    series = sampled.to(device)
    ctx = model.native_context()

    def step(seed):
        _, _, mpv, pm = model.sample_windows(series, n_windows, 0, CFG["window_step"], scale_input=True, seed=seed, summary=True, want_samples=False)
        return torch.stack([mpv, pm]).cpu()

    step(1)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(3):
        curve = step(10 + i)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        ctx.nst_forward(series, n_windows, 0, CFG["window_step"], model.scaler_mean, model.scaler_std, gemm=model.cond_pred_model.gemm_mode)
    e1.record()
    torch.cuda.synchronize(device)
    fx_ms = e0.elapsed_time(e1) / 3
    S, Ld, D, dff = CFG["windows"], CFG["windows"] // 2 + CFG["pred_len"], 512, 256
    enc = 2 * S * D * 3 * D + 4 * S * S * D + 2 * S * D * D + 4 * S * D * dff
    dec = (2 * Ld * D * 3 * D + 2 * Ld * Ld * D + 2 * Ld * D * D) + (2 * Ld * D * D + 2 * S * D * 2 * D + 4 * Ld * S * D + 2 * Ld * D * D) + 4 * Ld * D * dff
    flop = (4 * enc + 2 * dec) * CFG["nodes"] * n_windows
    return {"value": n_windows / dt, "unit": "windows/s", "ms_per_step": dt * 1e3, "finite": bool(torch.isfinite(curve).all()),
            "fx_ms_per_step": fx_ms, "fx_gflop_per_node": (4 * enc + 2 * dec) / 1e9, "fx_tflops_fp32_equivalent": flop / fx_ms / 1e9,
            "fx_gemm": model.cond_pred_model.gemm_mode,
            "fx_frac": flop / fx_ms / 1e9 / (FP32_MATRIX_PEAK_TFLOPS if model.cond_pred_model.gemm_mode == "f32" else F16_MATRIX_PEAK_TFLOPS / 3.0),
            "fx_peak": "fp32 matrix pipe 157.3 TFLOP/s" if model.cond_pred_model.gemm_mode == "f32" else
                       "dense f16 MFMA peak 2500 TFLOP/s / 3 products per algorithmic product (split-f16 GEMMs)",
            "fx_vs_fp32_matrix_peak": flop / fx_ms / 1e9 / FP32_MATRIX_PEAK_TFLOPS,
            "parity": "f(x) layers restated from the published ns-Transformer (torch-timeseries absent): parity unpinned"}


def trained_checkpoint_records(device):
    """Secondary records on TRAINED weights: both shipped sampler checkpoints (T = 20; committed as data under tests/golden/)
    on a C2-like batch (100 nodes, N = 32), with the device counter of reverse steps that failed the split-f16 kernel's
    end-of-step guard and were re-run through the exact routine."""
    from mpv_amd.models.models import diffusion_models
    recs = {}
    for tag, nf, rolling in (("wofx", 2, 100), ("sim", 1, 50)):
        path = ROOT / "tests" / "golden" / ("ckpt_%s.npz" % tag)
        if not path.exists():
            continue
        z = np.load(path)
        sd = {k: torch.from_numpy(z[k]) for k in z.files}
        net = dict(device=device, scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=rolling,
                   diffusion_schedule="linear", diffusion_steps=20, beta_start=1e-4, beta_end=0.02, n_z_samples=32,
                   parallel_sample=32, windows=200, pred_len=200, dataset_nf=nf)
        m = diffusion_models("NsDiff_model_variants", net, train_model_select="cond_var").to(device)
        m.load_state_dict({k: sd[k] for k in m.state_dict()}, strict=True)
        m.eval()
        g = torch.Generator().manual_seed(5)
        W = 16
        series = (torch.cumsum(torch.randn(100, 200 + 5 * (W - 1), nf, generator=g) * 0.1, dim=1)).to(device)
        c = m.native_context()
        m.sample_windows(series, W, 0, 5, seed=1, summary=True, want_samples=False)
        c.chain_stats(reset=True)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(3):
            _, _, mpv, _ = m.sample_windows(series, W, 0, 5, seed=2 + i, summary=True, want_samples=False)
        mpv = mpv.cpu()
        dt = time.perf_counter() - t0
        steps, safe = c.chain_stats()
        recs[tag] = {"checkpoint": "ews_results/%s" % ("NsDiff_machine/wo_fx" if tag == "wofx" else "real_data/model/simulation"),
                     "shape": "100 nodes, F=%d, windows=pred_len=200, N=32, T=20, %d windows per launch" % (nf, W),
                     "value": 3 * W / dt, "unit": "windows/s", "tile_steps": steps, "safe_steps": safe,
                     "finite": bool(torch.isfinite(mpv).all())}
        log("trained checkpoint %s: %.1f windows/s, %d of %d tile-steps through the exact routine" % (tag, recs[tag]["value"], safe, steps))
    return recs


DTYPES = {"f32": "f32",
          "f16x3": "f32 (128x128 layers as 3x f16 MFMA with fp32 accumulate)",
          "f16x2": "f32 elementwise + accumulate; 128x128 layers with two-term f16 weights and ONE-term f16 activations (2 MFMA products)",
          "f16": "f32 elementwise + accumulate; 128x128 layers with f16 weights and f16 activations (1 MFMA product)"}


def measured_traffic(n_windows, store_samples=False):
    """HBM bytes per launch of the headline chain kernel from the committed PMC passes of THIS binary (profiles/*/pmc_traffic_c2.json, written
    by tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE passes with the guide's gfx950 corrections, carrying the sha256 of the library
    it was measured on); None when there is no pass for this workload, `stale: True` when the pass belongs to another binary."""
    from mpv_amd.bench_common import measured_traffic as mt
    tr = mt("c2" + ("_store" if store_samples else ""))
    if not tr or tr.get("windows_in_step") != n_windows:
        return None
    for name, rec in (tr.get("kernels") or {}).items():
        if name.startswith("nsdiff_chain_f16_kernel"):
            per_launch = rec["hbm_bytes"] / max(int(rec.get("launches", 1)), 1)
            algorithmic = n_windows * CFG["nodes"] * CFG["pred_len"] * CFG["nf"] * 4 * (1 + (CFG["n_z"] if store_samples else 0)) + 256 * 150 * 1024
            return {"hbm_bytes_per_launch": None if tr["stale"] else per_launch, "stale": tr["stale"], "measured_hbm_bytes_per_launch": per_launch,
                    "algorithmic_bytes_per_launch": algorithmic, "source": tr["source"], "lib_sha256": tr.get("_lib_sha256"), "commit": tr.get("_commit")}
    return None


def roofline(gemm, achieved, kernel_ms, launches, flop_per_launch, n_windows, store_samples=False):
    """`achieved` = algorithmic TFLOP/s of the chain kernel (256*(256+5F) flop per chain-step)."""
    if gemm == "f32":
        return {"bound": "mfma", "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_MATRIX_PEAK_TFLOPS, "traffic": None, "kernel": "nsdiff_chain_kernel<1,true>",
                "kernel_avg_ms": kernel_ms, "launches": launches, "flop_per_launch": flop_per_launch,
                "peak_source": "fp32 matrix pipe (v_mfma_f32_32x32x2_f32)"}
    # split-f16: three f16 MFMA flops are executed per algorithmic flop of the two 128x128 layers, so the
    # matrix-pipe ceiling for ALGORITHMIC flops is the dense f16 peak / 3
    products = {"f16x3": 3, "f16x2": 2, "f16": 1}[gemm]
    if products != 3:
        peak = F16_MATRIX_PEAK_TFLOPS / products
        return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                "kernel": "nsdiff_chain_f16_kernel<1,true,%d>" % products, "kernel_avg_ms": kernel_ms, "launches": launches,
                "flop_per_launch": flop_per_launch,
                "peak_source": "dense f16 MFMA peak 2500 TFLOP/s / %d MFMA products per algorithmic product" % products,
                "note": "reduced-precision operand mode; live bound is vector-instruction issue (DESIGN.md 3.1)"}
    peak = F16_MATRIX_PEAK_TFLOPS / 3.0
    tr = measured_traffic(n_windows, store_samples)
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": (tr or {}).get("hbm_bytes_per_launch"), "traffic_stale": bool((tr or {}).get("stale")), "traffic_detail": tr,
            "kernel": "nsdiff_chain_f16_kernel<1,true>",
            "kernel_avg_ms": kernel_ms, "launches": launches,
            "flop_per_launch": flop_per_launch,
            "peak_source": "dense f16 MFMA peak 2500 TFLOP/s / 3 MFMA products per algorithmic product (f16x3 split)",
            "mfma_executed_tflops": 3.0 * achieved * 65536.0 / 66816.0,
            "vs_fp32_matrix_peak": achieved / FP32_MATRIX_PEAK_TFLOPS,
            "note": "live bound is vector-instruction issue under a power limit (12.2 k SIMD cycles per 32-chain step against 6.8 k of matrix "
                    "pipe, VALU issue active 89 %, shader clock 1.87 GHz under load: profiles/r03/pmc_util_f16x3.json), not the matrix pipe: "
                    "DESIGN.md 3.1"}


def cpu_baseline_and_parity(model, sampled, n_cpu_windows, modes):
    """Time the CPU oracle on the first window(s) of the same workload and compare MPV on shared noise, for every
    GEMM mode in `modes`."""
    from oracle import nsdiff_oracle as O
    cores = cpu_share()
    torch.set_num_threads(cores)
    log("cpu baseline: %d window(s) on %d threads" % (n_cpu_windows, cores))
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.NsDiffConfig(CFG["windows"], CFG["pred_len"], CFG["nf"], CFG["rolling"], CFG["steps"],
                         n_z_samples=CFG["n_z"], parallel_sample=CFG["parallel_sample"])
    wins = [sampled[:, w * CFG["window_step"]: w * CFG["window_step"] + CFG["windows"], :] for w in range(n_cpu_windows)]
    rec, cpu_mpv, cpu_out, per_window_s = [], [], [], []
    torch.manual_seed(2024)
    t0 = time.perf_counter()
    with torch.no_grad():
        for w in wins:
            t1 = time.perf_counter()
            x = O.scale(w, sd["scaler_mean"], sd["scaler_std"])
            outs = O.evaluation_step(sd, cfg, x, record=rec)
            cpu_out.append(outs)
            cpu_mpv.append(O.mpv_network(outs, sd["scaler_mean"], sd["scaler_std"])[1])
            per_window_s.append(time.perf_counter() - t1)
            log("cpu baseline window %d: %.1f s" % (len(per_window_s), per_window_s[-1]))
    cpu_s = time.perf_counter() - t0
    log("cpu baseline done in %.1f s" % cpu_s)
    # same windows on the GPU with the very noise the CPU run consumed (reference draw order)
    per = len(rec) // n_cpu_windows
    ctx = model.native_context()
    cpu_mpv = np.array(cpu_mpv)
    per_mode = {}
    for gemm in modes:
        model.gemm_mode = gemm
        gpu_mpv, samp_err = [], 0.0
        for i, w in enumerate(wins):
            noise = torch.cat([z.reshape(-1) for z in rec[i * per:(i + 1) * per]])
            x = O.scale(w, sd["scaler_mean"], sd["scaler_std"])
            outs, _ = model.evaluation_step(x, noise=noise)
            samp_err = max(samp_err, float((outs - cpu_out[i]).abs().max() / cpu_out[i].abs().max()))
            # the MPV as the bench step forms it: fused into the sampler's epilogue, same injected noise
            z6 = noise.reshape(1, -1, CFG["steps"], x.shape[0] * CFG["parallel_sample"], CFG["pred_len"], CFG["nf"])
            _, _, mpv, _ = model.sample_windows(x, 1, 0, 1, noise=z6, summary=True, want_samples=False)
            gpu_mpv.append(float(mpv[0]))
        gpu_mpv = np.array(gpu_mpv)
        ulp = np.spacing(np.abs(cpu_mpv).astype(np.float32)).astype(np.float64)           # the curve is float32: its spacing at each window's MPV
        per_mode[gemm] = {"mpv_max_abs_err": float(np.abs(gpu_mpv - cpu_mpv).max()),
                          "mpv_max_abs_err_ulps": float((np.abs(gpu_mpv - cpu_mpv) / ulp).max()),
                          "mpv_max_rel_err": float((np.abs(gpu_mpv - cpu_mpv) / np.abs(cpu_mpv)).max()),
                          "samples_max_rel_err": samp_err}
    model.gemm_mode = modes[0]
    return {
        "cpu_baseline": {"value": n_cpu_windows / cpu_s, "unit": "windows/s", "cores": cores, "kind": "port",
                         "best_window_value": 1.0 / min(per_window_s), "per_window_s": [round(t, 2) for t in per_window_s],
                         "sample": "first {} of the 181 windows, full N=32 x T=50, oracle/nsdiff_oracle.py "
                                   "(torch CPU fp32 port of the reference eager path), {:.1f} s in all; value = windows / total, "
                                   "best_window_value = 1 / fastest window".format(n_cpu_windows, cpu_s)},
        "per_mode": per_mode,
    }


if __name__ == "__main__":
    main()
