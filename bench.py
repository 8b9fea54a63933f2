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

Prints ONE JSON line (rank 0): metric/value/unit as BASELINE.json names them, plus
  roofline     -- the chain kernel against the fp32 matrix-pipe peak, timed live with HIP events
  cpu_baseline -- the CPU oracle (a torch-CPU port of the reference path) on a bounded sample
  mpv_*_err    -- MPV of the HIP path vs that CPU run on the SAME noise
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
    """Usable host cores: the cgroup CPU quota when there is one, else the affinity mask."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(cores, int(os.environ.get("MPV_CPU_THREADS", cores))))


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-windows", type=int, default=1, help="windows of the workload timed on the CPU oracle")
    ap.add_argument("--gemm", choices=("f16x3", "f32", "f16x2", "f16"), default="f16x3",
                    help="evaluation of the two 128x128 layers for the HEADLINE number: split-f16 matrix cores, fp32-equivalent "
                         "(default), exact fp32 MFMA, or the reduced-precision operand modes f16x2 / f16 (fp32 accumulate)")
    ap.add_argument("--host-data", action="store_true",
                    help="generate the synthetic trajectory with the numpy Euler-Maruyama on the host instead of the HIP kernel")
    ap.add_argument("--no-other-modes", action="store_true",
                    help="skip the extra, separately reported runs of the reduced-precision operand modes (N=1 only)")
    ap.add_argument("--max-windows", type=int, default=None, help="profiling aid: cut the trajectory to this many windows "
                    "(the reported metric is then NOT the BASELINE workload and says so)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE={}: launch with torch.distributed.run".format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("MPV_BENCH_FORCE_DIST") == "1"     # force: smoke the RCCL path on one GPU
    if use_dist:
        if os.environ.get("NCCL_DEBUG", "").upper() == "VERSION":    # keeps RCCL's version banner off stdout: rank 0 prints
            del os.environ["NCCL_DEBUG"]                             # exactly one line there, the JSON record
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

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

    def step(seed, gemm=args.gemm):
        gx = ctx.sigma_head(series_dev, n_windows, 0, CFG["window_step"], mean_dev, std_dev, add_eps=model.EPS)
        samples = ctx.sample(gx, None, S, P, seed=seed, gemm=gemm)
        mpv, pm, _ = ctx.reduce(samples, std_dev, mean_dev)
        curve = torch.stack([mpv, pm])                                                      # [2, W]
        if use_dist:
            out = torch.empty(world, 2, n_windows, device=device)
            dist.all_gather_into_tensor(out, curve)
            curve = out
        return curve.cpu()                                                                   # host copy = sync

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    log("%d windows resident; warmup" % n_windows)
    for i in range(args.warmup):
        step(i)
        log("warmup step %d done" % i)
    ctx.timing_enable(True)
    ctx.timing()                                                                             # drop warmup records
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        curve = step(100 + i)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.timing()
    ctx.timing_enable(False)
    log("timed %d steps in %.3f s; chain kernel avg %.2f ms" % (args.steps, elapsed, kernel_ms))
    if use_dist:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
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
                   "weights": "seeded default init (shipped checkpoints pin T=20)"},
        "roofline": roofline(args.gemm, achieved, kernel_ms, launches, flop_per_launch, n_windows),
    }

    # ---- reduced-precision operand modes, reported separately; never the headline unless --gemm asks
    others = [] if (world > 1 or args.no_other_modes) else [g for g in ("f16x2", "f16") if g != args.gemm]
    other_modes = {}
    for g in others:
        n_steps = max(1, min(args.steps, 3))
        step(7, g)
        ctx.timing_enable(True)
        ctx.timing()
        fence()
        t0 = time.perf_counter()
        for i in range(n_steps):
            step(200 + i, g)
        fence()
        dt = time.perf_counter() - t0
        kms, _ = ctx.timing()
        ctx.timing_enable(False)
        products = {"f16x2": 2, "f16": 1}[g]
        other_modes[g] = {"value": n_windows * n_steps / dt, "unit": "windows/s", "steps": n_steps, "ms_per_step": dt / n_steps * 1e3,
                          "dtype": DTYPES[g], "kernel_avg_ms": kms,
                          "achieved_tflops": flop_per_launch / (kms * 1e-3) / 1e12 if kms > 0 else 0.0,
                          "mfma_peak_tflops": F16_MATRIX_PEAK_TFLOPS / products}
        log("mode %s: %.1f windows/s" % (g, other_modes[g]["value"]))
    if other_modes:
        out["other_modes"] = other_modes

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        parity = cpu_baseline_and_parity(model, sampled, args.cpu_windows, [args.gemm] + others)
        for g, err in parity.pop("per_mode").items():
            if g == args.gemm:
                out.update(err)
            else:
                other_modes[g].update(err)
        out.update(parity)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


DTYPES = {"f32": "f32",
          "f16x3": "f32 (128x128 layers as 3x f16 MFMA with fp32 accumulate)",
          "f16x2": "f32 elementwise + accumulate; 128x128 layers with two-term f16 weights and ONE-term f16 activations (2 MFMA products)",
          "f16": "f32 elementwise + accumulate; 128x128 layers with f16 weights and f16 activations (1 MFMA product)"}


def measured_traffic(kernel, n_windows):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/*/pmc_traffic*.json, collected by
    tools/run_pmc_traffic.sh with the guide's gfx950 corrections); None when no pass matches this workload."""
    import glob
    for f in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "pmc_traffic*.json")), reverse=True):
        try:
            with open(f) as fh:
                rec = json.load(fh).get(kernel)
        except (OSError, ValueError):
            continue
        if rec and rec.get("windows") == n_windows:
            return {"hbm_bytes_per_launch": rec["hbm_bytes_per_launch"], "algorithmic_bytes_per_launch": rec["algorithmic_bytes_per_launch"],
                    "source": os.path.relpath(f, os.path.dirname(os.path.abspath(__file__)))}
    return None


def roofline(gemm, achieved, kernel_ms, launches, flop_per_launch, n_windows):
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
    tr = measured_traffic("nsdiff_chain_f16_kernel<1,true>", n_windows)
    return {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": (tr or {}).get("hbm_bytes_per_launch"), "traffic_detail": tr, "kernel": "nsdiff_chain_f16_kernel<1,true>",
            "kernel_avg_ms": kernel_ms, "launches": launches,
            "flop_per_launch": flop_per_launch,
            "peak_source": "dense f16 MFMA peak 2500 TFLOP/s / 3 MFMA products per algorithmic product (f16x3 split)",
            "mfma_executed_tflops": 3.0 * achieved * 65536.0 / 66816.0,
            "vs_fp32_matrix_peak": achieved / FP32_MATRIX_PEAK_TFLOPS,
            "note": "live bound is vector-instruction issue (about 12.6 k issue cycles per 32-chain step against 6.5 k of matrix "
                    "pipe, shader clock 1.9 GHz under load), not the matrix pipe: DESIGN.md 3.1 item 3"}


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
    rec, cpu_mpv, cpu_out = [], [], []
    torch.manual_seed(2024)
    t0 = time.perf_counter()
    with torch.no_grad():
        for w in wins:
            x = O.scale(w, sd["scaler_mean"], sd["scaler_std"])
            outs = O.evaluation_step(sd, cfg, x, record=rec)
            cpu_out.append(outs)
            cpu_mpv.append(O.mpv_network(outs, sd["scaler_mean"], sd["scaler_std"])[1])
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
            mpv, _, _ = ctx.reduce(outs.to(ctx.device).unsqueeze(0).contiguous(), model.scaler_std.float().contiguous(),
                                   model.scaler_mean.float().contiguous())
            gpu_mpv.append(float(mpv[0]))
        gpu_mpv = np.array(gpu_mpv)
        per_mode[gemm] = {"mpv_max_abs_err": float(np.abs(gpu_mpv - cpu_mpv).max()),
                          "mpv_max_rel_err": float((np.abs(gpu_mpv - cpu_mpv) / np.abs(cpu_mpv)).max()),
                          "samples_max_rel_err": samp_err}
    model.gemm_mode = modes[0]
    return {
        "cpu_baseline": {"value": n_cpu_windows / cpu_s, "unit": "windows/s", "cores": cores, "kind": "port",
                         "sample": "first {} of the 181 windows, full N=32 x T=50, oracle/nsdiff_oracle.py "
                                   "(torch CPU fp32 port of the reference eager path), {:.1f} s".format(n_cpu_windows, cpu_s)},
        "per_mode": per_mode,
    }


if __name__ == "__main__":
    main()
