"""bench.py --config c5: DiffSTG (BASELINE configs[4]) on a 200-node small-world biomass trajectory, N = 32 samples, 50 inference steps.

One "step" = `--stg-windows` rolling windows (default 4) in one library call: every window is one evaluation_step of the reference
with parallel_sampling = 32 graph copies (6400 node rows of [T = 200] per window) through 50 DDIM steps of the 12-residual-block
graph U-Net, then the MPV reduction of the last pred_len positions.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np
import torch

C5 = dict(nodes=200, ticks=20_000, sampling_t=1, windows=100, pred_len=100, window_step=50, parallel=32, sequential=1, steps=50,
          d_h=4, Td_h=10, channel_multipliers=[2, 2], n_blocks=2, diffusion_steps=200)


def small_world(n, k=4, p=0.1, seed=0):
    """Watts-Strogatz ring lattice with rewiring (symmetric 0/1 adjacency)."""
    rng = np.random.default_rng(seed)
    adj = np.zeros((n, n))
    for i in range(n):
        for j in range(1, k // 2 + 1):
            a, b = i, (i + j) % n
            if rng.random() < p:
                b = int(rng.integers(n))
                while b == a or adj[a, b]:
                    b = int(rng.integers(n))
            adj[a, b] = adj[b, a] = 1.0
    return adj


def flops_per_row(model):
    """Algorithmic FLOPs (2 per multiply-add) of one UGnet forward for one graph-node row: TCN convolutions, k = T+1 down / up
    convolutions, the four Linear maps of the gated graph convolution, 1x1 shortcuts, resampling convolutions, output head."""
    tf = model.model
    td = int(tf.net_param["Td_h"])
    total, T_cur = 0, 2 * tf.T
    for st in tf.layout():
        m = st["module"]
        if st["kind"] == "res":
            ci, co, T = m.c_in, m.c_out, m.T
            C = td * co
            total += 2 * co * ci * 3 * T + 2 * co * co * 3 * T + 2 * co * td * co * (T + 1) * 2 + 2 * 4 * C * C
            total += 2 * co * ci * T * (2 if ci != co else 0)
        else:
            c = m.conv.weight.shape[0]
            total += 2 * c * c * (3 * T_cur // 2 if st["kind"] == "down" else 4 * T_cur)
            T_cur = (T_cur - 1) // 2 + 1 if st["kind"] == "down" else 2 * T_cur
    total += 2 * tf.d_h * tf.F * 2 * tf.T * 2 + 2 * 2 * tf.T * tf.T * tf.F
    return total


def main_c5(args, log):
    from mpv_amd.bench_common import Ranks
    rk = Ranks(args)                                   # N ranks: every rank samples windows of its own trajectory (weak scaling), graph replicated
    out = record_c5(args, log, rk)
    if rk.rank == 0:
        from mpv_amd.bench_common import emit_record, print_line
        emit_record("c5", out, fresh=True)
        print_line(out)
    rk.close()


def record_c5(args, log, rk):
    """One bench record of BASELINE configs[4] (dict; printed by `--config c5`, embedded as other_configs.c5 by the default run)."""
    from mpv_amd import synthetic
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.bench_common import measured_traffic, cpu_share, multi_kernel_roofline
    from mpv_amd.bench_dts import slim_traffic
    device = rk.device
    cfg = C5
    from mpv_amd import _native
    adj = small_world(cfg["nodes"], seed=3000)
    row_ptr, col_idx = synthetic.adjacency_csr(adj)
    stride = H.sampling_interval_from_t(cfg["sampling_t"])
    gen = _native.Context(device)
    x0 = np.random.default_rng(3000).uniform(0.0, 1.0, cfg["nodes"]).astype(np.float32)
    series = gen.biomass_simulate(torch.from_numpy(row_ptr).to(device), torch.from_numpy(col_idx).to(device), torch.from_numpy(x0).to(device).unsqueeze(0),
                                  cfg["ticks"], 1000, stride, 0.7, 10.0, 0.5, 0.005, 0.1, 0.0, 2.7, 3000 + rk.rank)[0]            # [V, L, 1]
    gen.close()
    log("c5: device biomass trajectory on a 200-node small-world graph: %s" % (tuple(series.shape),))
    n_total = H.sliding_window_count(series.shape[1], cfg["windows"], cfg["window_step"])
    wps = max(1, int(getattr(args, "stg_windows", 4)))
    net = dict(device=device, scaler_type="StandardScaler", task_model="DiffSTG", F=1, T_h=cfg["windows"], T_p=cfg["pred_len"], Td_h=cfg["Td_h"],
               d_h=cfg["d_h"], channel_multipliers=cfg["channel_multipliers"], n_blocks=cfg["n_blocks"], diffusion_steps=cfg["diffusion_steps"],
               inference_diffusion_steps=cfg["steps"], inference_schedule="cosine", inference_trick="ddim", diffusion_schedule="linear",
               loss_weight_schedule="constant", mask_ratio=0.0, parallel_sampling=cfg["parallel"], sequential_sampling=cfg["sequential"],
               gnn_name="ResGatedGraphConv", gnn_param={"bias": True, "root_weight": True})
    torch.manual_seed(321)
    model = diffusion_models("DiffSTG", net).to(device).eval()
    flat = series.reshape(-1, 1).float()
    model.scaler_mean, model.scaler_std = flat.mean(dim=0), flat.std(dim=0)
    model.refresh_native()
    ctx = model.native_context()
    src, dst = np.nonzero(adj)
    edge_index = torch.from_numpy(np.stack([src, dst])).long()
    scaled = ((series - model.scaler_mean) / model.scaler_std).contiguous()
    std_dev, mean_dev = model.scaler_std.float().contiguous(), model.scaler_mean.float().contiguous()

    def step(i, seed):
        w0 = (i * wps) % max(1, n_total - wps + 1)
        idx = (torch.arange(wps, device=device) + w0).view(-1, 1) * cfg["window_step"] + torch.arange(cfg["windows"], device=device)
        hist = scaled[:, idx, :].permute(1, 0, 2, 3).contiguous()                                  # [W, V, T_h, 1]
        samples = model.sample_windows(hist, edge_index, seed=seed)[:, :, -cfg["pred_len"]:].contiguous()
        mpv, pm, _ = ctx.reduce(samples, std_dev, mean_dev)
        return rk.gather(torch.stack([mpv, pm]))

    for i in range(args.warmup):
        step(i, i)
        log("c5 warmup step %d done" % i)
    rk.fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        curve = step(i, 100 + i + 1000 * rk.rank)
    rk.fence()
    elapsed = rk.max_time(time.perf_counter() - t0)
    assert bool(torch.isfinite(curve).all())
    rows = cfg["nodes"] * cfg["parallel"] * cfg["sequential"]
    fpr = flops_per_row(model)
    flop_per_window = rows * cfg["steps"] * fpr
    achieved = flop_per_window * wps * args.steps / elapsed / 1e12
    tr = measured_traffic("c5")
    out = {
        "metric": "rolling windows/sec (N=32, 50 inference steps), DiffSTG", "value": rk.world * wps * args.steps / elapsed, "unit": "windows/s", "n_gpus": rk.world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 (k = T+1 convolutions, graph-conv Linear maps and TCN blocks as two-term f16 splits on the f16 matrix pipe: 3 MFMA products per algorithmic product, fp32 accumulate; rest fp32 VALU)", "data": "synthetic",
        "config": {"workload": "DiffSTG on 200-node small-world biomass trajectory (BASELINE configs[4]): T_h=T_p=100, d_h=4, multipliers [2,2], "
                               "2 blocks per resolution (12 residual blocks), Td_h=10, parallel_sampling=32 (6400 node rows per window), 50 DDIM steps; "
                               "{} window(s) per step of the {} of the trajectory".format(wps, n_total),
                   "windows_per_step_per_gpu": wps, "noise": "on-device Philox4x32-10", "weights": "seeded default init"},
        "roofline": multi_kernel_roofline(
            tr, rk.world * wps * args.steps / elapsed / rk.world, achieved, "f16x3",     # bench_stg builds the model with the default gemm_mode
            "nst_gemm_h3_kernel (k = T+1 convolutions, graph-conv Linear maps) + stg_tcn_mfma_kernel (TCN blocks) + stg_gate_kernel",
            {"flop_per_window": flop_per_window, "flop_per_row_forward": fpr, "traffic_detail": slim_traffic(tr),
             "note": "60 kernels per inference step over [6400 rows x c x T] activations: every kernel is bound by the activation traffic it streams "
                     "(GEMMs 3 TB/s, TCN blocks 2.4-3.2 TB/s), not by either arithmetic pipe"}),
    }
    if not args.no_cpu_baseline and rk.world == 1:
        from oracle import diffstg_oracle as O
        cores = cpu_share()
        torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        onet = dict(net, parallel_sampling=4, sequential_sampling=1)
        g = torch.Generator().manual_seed(1)
        log("c5 cpu baseline: 4 of the 32 graph copies of one window, all 50 steps, on %d threads" % cores)
        t0 = time.perf_counter()
        outs = O.evaluation_step(sd, onet, scaled[:, : cfg["windows"], :].cpu(), edge_index, lambda shape: torch.randn(shape, generator=g))
        dt = time.perf_counter() - t0
        share = 4.0 / 32.0
        out["cpu_baseline"] = {"value": share / dt, "unit": "windows/s", "cores": cores, "kind": "port", "finite": bool(torch.isfinite(outs).all()),
                               "sample": "4 of the 32 graph copies (800 node rows) x all 50 inference steps = 1/8 of a window, oracle/diffstg_oracle.py "
                                         "(torch CPU fp32 port), {:.1f} s; value = (1/8 window) / time".format(dt)}
    return out
