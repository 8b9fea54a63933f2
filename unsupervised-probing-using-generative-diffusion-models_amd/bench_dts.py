"""bench.py --config c3: DiffusionTS (BASELINE configs[2]) on a 100-node SIS trajectory, N = 32 samples, T = 100 sampling steps.

One "step" = `--dts-windows` rolling windows (default 1): every window is one evaluation_step of the reference = 3200 sequences
(100 nodes x 32 samples, all in one chunk) of 200 tokens through 100 DDIM steps with the Langevin infill (100 forward passes +
128 forward/backward passes of the 3 + 6 block transformer), then the MPV reduction.  Series and weights are resident in HBM.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

C3 = dict(nodes=100, ticks=10_000, sampling_t=0.1, windows=100, pred_len=100, window_step=50, n_z=32, parallel_sample=32, nf=1,
          d_model=64, n_heads=4, n_layer_enc=3, n_layer_dec=6, timesteps=100, diffusion_steps=100)


def flops_per_sequence(L, D, H, M, n_enc, n_dec, NF):
    """Algorithmic FLOPs (2 per multiply-add) of one transformer forward and of its input-gradient pass, dense contractions only."""
    attn_f = 2 * L * D * 3 * D + 2 * 2 * L * L * D + 2 * L * D * D           # QKV, scores + PV, proj
    attn_b = 2 * L * D * 3 * D + 4 * 2 * L * L * D + 2 * L * D * D           # transposed linears; dV, dP, dQ, dK
    mlp = 2 * 2 * L * D * M
    enc_f, enc_b = attn_f + mlp, attn_b + mlp
    mix = 2 * (2 * L) * L * D
    dft = 2 * (2 * NF) * L * D
    dec_f = 2 * attn_f + mlp + mix + dft
    dec_b = 2 * attn_b + mlp + mix
    return n_enc * enc_f + n_dec * dec_f, n_enc * enc_b + n_dec * dec_b


def main_c3(args, log):
    from mpv_amd.bench_common import Ranks
    rk = Ranks(args)                                   # N ranks: every rank samples windows of its own trajectory (weak scaling)
    out = record_c3(args, log, rk)
    if rk.rank == 0:
        from mpv_amd.bench_common import emit_record, print_line
        emit_record("c3", out, fresh=True)             # the full record (per-kernel traffic table, notes): stderr + records file
        print_line(out)
    rk.close()


def record_c3(args, log, rk):
    """One bench record of BASELINE configs[2] (dict; printed by `--config c3`, embedded as other_configs.c3 by the default run)."""
    from mpv_amd import synthetic
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.bench_common import measured_traffic, multi_kernel_roofline
    device = rk.device
    cfg = dict(C3)
    short = int(os.environ.get("MPV_BENCH_DTS_STEPS", "0"))              # profiling aid (PMC passes): fewer DDIM steps, flagged in the record
    if short > 0:
        cfg["diffusion_steps"] = short
    from mpv_amd import _native
    gen = _native.Context(device)
    sampled, _ = synthetic.sis_trajectory_device(gen, cfg["nodes"], cfg["ticks"], seed=2000 + rk.rank, sampling_stride=H.sampling_interval_from_t(cfg["sampling_t"]))
    sampled = sampled.cpu()
    gen.close()
    log("c3: SIS trajectory generated on device: %s" % (tuple(sampled.shape),))
    n_total = H.sliding_window_count(sampled.shape[1], cfg["windows"], cfg["window_step"])
    wps = max(1, int(getattr(args, "dts_windows", 1)))
    net = dict(device=device, scaler_type="StandardScaler", task_model="DiffusionTS", dataset_nf=cfg["nf"], windows=cfg["windows"],
               pred_len=cfg["pred_len"], d_model=cfg["d_model"], n_heads=cfg["n_heads"], n_layer_enc=cfg["n_layer_enc"],
               n_layer_dec=cfg["n_layer_dec"], timesteps=cfg["timesteps"], diffusion_steps=cfg["diffusion_steps"],
               n_z_samples=cfg["n_z"], parallel_sample=cfg["parallel_sample"], gemm_mode=os.environ.get("MPV_DTS_GEMM", "f16x3"))
    torch.manual_seed(123)
    model = diffusion_models("DiffusionTS", net).to(device).eval()
    flat = sampled.reshape(-1, cfg["nf"]).float()
    model.scaler_mean, model.scaler_std = flat.mean(dim=0).to(device), flat.std(dim=0).to(device)
    ctx = model.native_context()
    series = ((sampled.to(device) - model.scaler_mean) / model.scaler_std).contiguous()          # [V, L, F] scaled, resident
    std_dev, mean_dev = model.scaler_std.float().contiguous(), model.scaler_mean.float().contiguous()

    def step(i, seed):
        w0 = (i * wps) % max(1, n_total - wps + 1)
        idx = (torch.arange(wps, device=device) + w0).view(-1, 1) * cfg["window_step"] + torch.arange(cfg["windows"], device=device)
        hist = series[:, idx, :].permute(1, 0, 2, 3).contiguous()                                 # [W, V, windows, F]
        samples = model.sample_windows(hist, seed=seed)
        mpv, pm, _ = ctx.reduce(samples, std_dev, mean_dev)
        return rk.gather(torch.stack([mpv, pm]))

    for i in range(args.warmup):
        step(i, i)
        log("c3 warmup step %d done" % i)
    rk.fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        curve = step(i, 100 + i + 1000 * rk.rank)
    rk.fence()
    elapsed = rk.max_time(time.perf_counter() - t0)
    assert bool(torch.isfinite(curve).all())
    L, D, Hh, M = cfg["windows"] + cfg["pred_len"], cfg["d_model"], cfg["n_heads"], 4 * cfg["d_model"]
    ff, fb = flops_per_sequence(L, D, Hh, M, cfg["n_layer_enc"], cfg["n_layer_dec"], L // 2 - 1)
    plan = model._plan()
    n_fwd = plan.shape[0] + int(plan[:, 6].sum())                      # model_predictions + one forward per Langevin iteration
    n_bwd = int(plan[:, 6].sum())
    seqs = cfg["nodes"] * cfg["n_z"]
    flop_per_window = seqs * (n_fwd * ff + n_bwd * fb)
    achieved = flop_per_window * wps * args.steps / elapsed / 1e12                       # per GPU
    tr = measured_traffic("c3")
    out = {
        "metric": "rolling windows/sec (N=32, T=100), DiffusionTS", "value": rk.world * wps * args.steps / elapsed, "unit": "windows/s", "n_gpus": rk.world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (every contraction on v_mfma_f32_16x16x4_f32)" if net["gemm_mode"] == "f32" else
        "f32 (attention, fused MLP, token-wise Linear and token-mixing products as two-term f16 splits on the f16 matrix pipe: 3 MFMA products per "
        "algorithmic product, fp32 accumulate, fp32-equivalent)", "data": "synthetic",
        "config": {"workload": "DiffusionTS on 100-node BA-graph SIS trajectory (BASELINE configs[2]): windows=100, pred_len=100, d_model=64, "
                               "4 heads, 3 enc + 6 dec blocks, N=32 samples in one chunk (3200 sequences x 200 tokens), T=100 DDIM steps with "
                               "Langevin infill; {} window(s) per step of the {} of the trajectory".format(wps, n_total),
                   "windows_per_step_per_gpu": wps, "noise": "on-device Philox4x32-10", "weights": "seeded default init",
                   "truncated_for_profiling": short > 0, "sampling_steps": cfg["diffusion_steps"],
                   "passes_per_window": {"forward": n_fwd, "backward": n_bwd}},
        "roofline": multi_kernel_roofline(
            tr if short == 0 else None, rk.world * wps * args.steps / elapsed / rk.world, achieved, "f32" if net["gemm_mode"] == "f32" else "f16x3",
            "dts_lin_wave + dts_attn_{fwd,dq,dkv}_h + dts_mlp_{fwd,bwd}_h + dts_ln_bwd + dts_seqmix + Fourier / trend kernels (whole sampler call)",
            {"flop_per_window": flop_per_window, "flop_per_sequence_forward": ff, "flop_per_sequence_backward": fb, "traffic_detail": slim_traffic(tr),
             "note": "356 transformer passes of about 100 kernels each over [3200 x 200 x 64] activations; achieved compute = algorithmic FLOPs of the "
                     "dense contractions / wall time of the timed steps"}),
    }
    if not args.no_cpu_baseline and rk.world == 1:
        out["cpu_baseline"] = cpu_baseline(model, series, cfg, log)
    return out


def slim_traffic(rec, top=8):
    """The committed PMC record without its long tail of tiny kernels (the JSON line stays readable)."""
    if not rec:
        return None
    out = {k: v for k, v in rec.items() if k != "kernels"}
    out["kernels"] = dict(list(rec.get("kernels", {}).items())[:top])
    return out


def cpu_baseline(model, series, cfg, log, nodes=4, par=4):
    """The CPU oracle (torch fp32 port of the reference sampler) on a bounded sample of the same workload: `nodes` x `par` sequences of
    the first window, all T = 100 steps; windows/s scaled by the share of the window's 3200 sequences that was computed.  16 sequences is
    where the per-sequence time stops falling with the batch (measured on 8 cores: 1.00 / 0.22 / 0.22 / 0.29 s per sequence at 4 / 16 /
    32 / 64 sequences), so the linear scaling to 3200 does not overstate the CPU cost."""
    from oracle import diffusionts_oracle as O
    from mpv_amd.bench_common import cpu_share
    cores = cpu_share()
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ocfg = O.DTSConfig(cfg["windows"], cfg["pred_len"], cfg["nf"], cfg["d_model"], cfg["n_layer_enc"], cfg["n_layer_dec"], cfg["n_heads"],
                       cfg["timesteps"], cfg["diffusion_steps"], n_z_samples=par, parallel_sample=par)
    batch = series[:nodes, : cfg["windows"], :].cpu()
    g = torch.Generator().manual_seed(1)
    n = nodes * par
    log("c3 cpu baseline: %d of the 3200 sequences of one window on %d threads" % (n, cores))
    t0 = time.perf_counter()
    outs = O.evaluation_step(sd, ocfg, batch, lambda shape: torch.randn(shape, generator=g))
    dt = time.perf_counter() - t0
    log("c3 cpu baseline done in %.1f s" % dt)
    share = float(n) / (cfg["nodes"] * cfg["n_z"])
    return {"value": share / dt, "unit": "windows/s", "cores": cores, "kind": "port", "finite": bool(torch.isfinite(outs).all()),
            "sample": "{} nodes x {} samples ({} of the 3200 sequences of one window: the batch at which the CPU's per-sequence time has "
                      "plateaued), all T={} steps, oracle/diffusionts_oracle.py (torch CPU fp32 port of the reference sampler incl. autograd "
                      "for the Langevin infill), {:.1f} s; value = ({}/3200 window) / time".format(nodes, par, n, cfg["diffusion_steps"], dt, n)}
