"""bench.py --config c4: NsDiff on a 500-node Erdos-Renyi Wilson-Cowan trajectory, N = 64 samples, T = 50 (BASELINE configs[3]).

STRONG scaling: ONE trajectory (181 rolling windows of 500 nodes) is split into contiguous window blocks over the ranks
(`distributed.block_bounds`, the harness's own partition), every rank runs sigma head + fused sampler / summary on its block, one all-gather
(RCCL; gloo for the rehearsal) reassembles the MPV curve on every rank.  value = windows of the whole trajectory per second.
"""
from __future__ import annotations

import json
import os
import time

import torch
import torch.distributed as dist

C4 = dict(nodes=500, ticks=100_000, sampling_t=10, windows=100, pred_len=100, rolling=50, window_step=5, steps=50, n_z=64, parallel_sample=64, nf=1)


def main_c4(args, log):
    from mpv_amd.bench_common import Ranks
    rk = Ranks(args)
    out = record_c4(args, log, rk)
    if rk.rank == 0:
        from mpv_amd.bench_common import emit_record, print_line
        emit_record("c4", out, fresh=True)
        print_line(out)
    rk.close()


def record_c4(args, log, rk):
    """One bench record of BASELINE configs[3] (dict; printed by `--config c4`, embedded as other_configs.c4 by the default run)."""
    import bench as B
    from mpv_amd import _native, synthetic
    from mpv_amd import distributed as D
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    cfg = C4
    world, rank, device = rk.world, rk.rank, rk.device
    stride = H.sampling_interval_from_t(cfg["sampling_t"])
    gen = _native.Context(device)
    series, _ = synthetic.neuronal_trajectory_device(gen, cfg["nodes"], cfg["ticks"], seed=4000, sampling_stride=stride)     # the SAME trajectory on every rank
    gen.close()
    sampled = series.cpu().contiguous()
    n_windows = H.sliding_window_count(sampled.shape[1], cfg["windows"], cfg["window_step"])
    cut = getattr(args, "max_windows", None)                 # profiling / test aid: the record says so (`truncated_for_profiling`)
    if cut is not None:
        n_windows = min(n_windows, int(cut))
    net_param = dict(device=device, scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=cfg["rolling"],
                     diffusion_schedule="linear", diffusion_steps=cfg["steps"], beta_start=1e-4, beta_end=0.02, n_z_samples=cfg["n_z"],
                     parallel_sample=cfg["parallel_sample"], windows=cfg["windows"], pred_len=cfg["pred_len"], dataset_nf=cfg["nf"])
    model = diffusion_models("NsDiff_model_variants", net_param, train_model_select="cond_var").to(device).eval()
    synthetic.seeded_nsdiff_state(model, sampled, seed=123)
    model.gemm_mode = args.gemm
    ctx = model.native_context()
    series_dev = sampled.to(device)
    mean_dev, std_dev = model.scaler_mean.float().contiguous(), model.scaler_std.float().contiguous()
    lo, hi = D.block_bounds(n_windows, rank, world)
    S, P = cfg["n_z"], cfg["parallel_sample"]
    log("c4 rank %d/%d: windows [%d, %d) of %d, %d nodes x %d samples" % (rank, world, lo, hi, n_windows, cfg["nodes"], S))

    def step(seed):
        if hi > lo:
            gx = ctx.sigma_head(series_dev, hi - lo, lo * cfg["window_step"], cfg["window_step"], mean_dev, std_dev, add_eps=model.EPS)
            _, mpv, pm, _ = ctx.sample(gx, None, S, P, seed=seed + 7919 * rank, gemm=args.gemm, summary=True, want_samples=False, std=std_dev, mean=mean_dev)
        else:
            mpv = pm = torch.zeros(0, device=device)
        return D.gather_curve(n_windows, lo, hi, mpv, pm)                     # every rank: the whole curve (host)

    fence = rk.fence

    for i in range(args.warmup):
        step(i)
    ctx.timing_enable(True)
    ctx.timing()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        mpv_curve, pm_curve = step(100 + i)
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.timing()
    ctx.timing_enable(False)
    elapsed = rk.max_time(elapsed)
    assert len(mpv_curve) == n_windows and bool(torch.isfinite(torch.as_tensor(mpv_curve)).all()), "bad MPV curve"
    flop_per_launch = (hi - lo) * cfg["nodes"] * cfg["pred_len"] * S * cfg["steps"] * 256 * (256 + 5 * cfg["nf"])
    achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    products = {"f16x3": 3, "f16x2": 2, "f16": 1, "f32": 0}[args.gemm]
    peak = B.FP32_MATRIX_PEAK_TFLOPS if args.gemm == "f32" else B.F16_MATRIX_PEAK_TFLOPS / products
    from mpv_amd.bench_common import measured_traffic
    tr = measured_traffic("c4") if world == 1 else None
    out = {
        "metric": "rolling windows/sec (N=64, T=50), NsDiff 500-node Wilson-Cowan", "value": n_windows * args.steps / elapsed, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": B.DTYPES[args.gemm], "data": "synthetic",
        "config": {"workload": "NsDiff cond_var on a 500-node Erdos-Renyi Wilson-Cowan trajectory (BASELINE configs[3]): windows=100, pred_len=100, "
                               "rolling=50, N=64 samples, T=50 steps, {} windows; ONE trajectory per step, its windows split in contiguous "
                               "blocks over {} rank(s), one all-gather of the curve".format(n_windows, world),
                   "windows_per_step": n_windows, "windows_of_rank0": hi - lo, "truncated_for_profiling": cut is not None, "noise": "on-device Philox4x32-10", "weights": "seeded default init",
                   "summary": "MPV reduction fused into the sampler epilogue, no sample tensor"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak if peak else None,
                     "traffic": None if (tr or {}).get("stale") else chain_traffic(tr), "traffic_stale": bool((tr or {}).get("stale")),
                     "traffic_detail": {k: v for k, v in (tr or {}).items() if k != "kernels"} or None,
                     "kernel": "nsdiff_chain_f16_kernel<1,true> (rank 0's block)", "kernel_avg_ms": kernel_ms, "launches": launches,
                     "flop_per_launch": flop_per_launch},
    }
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(model, sampled, cfg, log)
    return out


def chain_traffic(tr):
    """HBM bytes of ONE chain-kernel launch (181 windows) out of the committed PMC pass (profiles/*/pmc_traffic_c4.json)."""
    for name, rec in ((tr or {}).get("kernels") or {}).items():
        if name.startswith("nsdiff_chain"):
            return rec["hbm_bytes"] / max(int(rec.get("launches", 1)), 1)
    return None


def cpu_baseline(model, sampled, cfg, log, nodes=25):
    """The CPU oracle on a bounded sample of the same workload: the first window, the first `nodes` of the 500 nodes (NsDiff treats nodes
    as independent batch rows), full N = 64 x T = 50; windows/s scaled by the node share.  Also the MPV of exactly those rows from the HIP
    path on the oracle's own noise."""
    from oracle import nsdiff_oracle as O
    from mpv_amd.bench_common import cpu_share
    cores = cpu_share()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ocfg = O.NsDiffConfig(cfg["windows"], cfg["pred_len"], cfg["nf"], cfg["rolling"], cfg["steps"], n_z_samples=cfg["n_z"], parallel_sample=cfg["parallel_sample"])
    x = O.scale(sampled[:nodes, : cfg["windows"], :], sd["scaler_mean"], sd["scaler_std"])
    rec = []
    torch.manual_seed(2025)
    log("c4 cpu baseline: %d of the %d nodes of one window on %d threads" % (nodes, cfg["nodes"], cores))
    t0 = time.perf_counter()
    with torch.no_grad():
        outs = O.evaluation_step(sd, ocfg, x, record=rec)
    dt = time.perf_counter() - t0
    cpu_mpv = O.mpv_network(outs, sd["scaler_mean"], sd["scaler_std"])[1]
    noise = torch.cat([z.reshape(-1) for z in rec])
    z6 = noise.reshape(1, -1, cfg["steps"], nodes * cfg["parallel_sample"], cfg["pred_len"], cfg["nf"])
    _, _, mpv, _ = model.sample_windows(x, 1, 0, 1, noise=z6, summary=True, want_samples=False)
    gpu_mpv = float(mpv[0])
    share = float(nodes) / cfg["nodes"]
    return {"value": share / dt, "unit": "windows/s", "cores": cores, "kind": "port", "mpv_cpu": float(cpu_mpv), "mpv_gpu_same_noise": gpu_mpv,
            "mpv_abs_err": abs(gpu_mpv - float(cpu_mpv)), "mpv_rel_err": abs(gpu_mpv - float(cpu_mpv)) / abs(float(cpu_mpv)),
            "sample": "first window, first {} of the {} nodes, full N={} x T={}, oracle/nsdiff_oracle.py (torch CPU fp32 port of the reference eager "
                      "path), {:.1f} s; value = ({}/{} window) / time".format(nodes, cfg["nodes"], cfg["n_z"], cfg["steps"], dt, nodes, cfg["nodes"])}
