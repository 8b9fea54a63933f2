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
    import bench as B
    from mpv_amd import _native, synthetic
    from mpv_amd import distributed as D
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    cfg = C4
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm device: the path has no CPU implementation")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    stride = H.sampling_interval_from_t(cfg["sampling_t"])
    gen = _native.Context(device)
    series, _ = synthetic.neuronal_trajectory_device(gen, cfg["nodes"], cfg["ticks"], seed=4000, sampling_stride=stride)     # the SAME trajectory on every rank
    gen.close()
    sampled = series.cpu().contiguous()
    n_windows = H.sliding_window_count(sampled.shape[1], cfg["windows"], cfg["window_step"])
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

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

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
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert len(mpv_curve) == n_windows and bool(torch.isfinite(torch.as_tensor(mpv_curve)).all()), "bad MPV curve"
    flop_per_launch = (hi - lo) * cfg["nodes"] * cfg["pred_len"] * S * cfg["steps"] * 256 * (256 + 5 * cfg["nf"])
    achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12 if kernel_ms > 0 else 0.0
    products = {"f16x3": 3, "f16x2": 2, "f16": 1, "f32": 0}[args.gemm]
    peak = B.FP32_MATRIX_PEAK_TFLOPS if args.gemm == "f32" else B.F16_MATRIX_PEAK_TFLOPS / products
    out = {
        "metric": "rolling windows/sec (N=64, T=50), NsDiff 500-node Wilson-Cowan", "value": n_windows * args.steps / elapsed, "unit": "windows/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": B.DTYPES[args.gemm], "data": "synthetic",
        "config": {"workload": "NsDiff cond_var on a 500-node Erdos-Renyi Wilson-Cowan trajectory (BASELINE configs[3]): windows=100, pred_len=100, "
                               "rolling=50, N=64 samples, T=50 steps, {} windows; ONE trajectory per step, its windows split in contiguous "
                               "blocks over {} rank(s), one all-gather of the curve".format(n_windows, world),
                   "windows_per_step": n_windows, "windows_of_rank0": hi - lo, "noise": "on-device Philox4x32-10", "weights": "seeded default init",
                   "summary": "MPV reduction fused into the sampler epilogue, no sample tensor"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak if peak else None, "traffic": None,
                     "kernel": "nsdiff_chain_f16_kernel<1,true> (rank 0's block)", "kernel_avg_ms": kernel_ms, "launches": launches,
                     "flop_per_launch": flop_per_launch},
    }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
