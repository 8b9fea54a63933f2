"""Secondary bench records of the two model families BASELINE.json's configs do not name: NsDiff_spatial and TMDM (row a18), on
C5- / C2-shaped synthetic workloads, each with a roofline object and a bounded CPU baseline (the oracle).  Embedded as `other_families` in the
default `python bench.py` line; never the headline."""
from __future__ import annotations

import os
import time

import numpy as np
import torch

F16_MATRIX_PEAK_TFLOPS = 2500.0

NSS = dict(nodes=200, windows=100, pred_len=100, rolling=50, steps=50, n_z=32, parallel=32, d_h=4, Td_h=10, channel_multipliers=[2, 2], n_blocks=2, n_windows=4)
TMDM = dict(nodes=100, windows=100, pred_len=100, steps=50, n_z=32, parallel=32, nf=1, n_windows=32)


def _events(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3, out


def record_nsdiff_spatial(device, log, cpu=True):
    """NsDiff_spatial on the C5 graph (200-node small world): windows of [200 nodes, windows 100 -> pred_len 100], N = 32 graph copies side by
    side, T = 50 reverse steps of the graph U-Net; f(x) supplied (seeded), g(x) by the sigma-head kernel."""
    from mpv_amd.bench_common import cpu_share
    from mpv_amd.bench_stg import small_world
    from mpv_amd.models.models import diffusion_models
    c = NSS
    net = dict(device=device, scaler_type="StandardScaler", task_model="NsDiff_spatial", dataset_nf=1, windows=c["windows"], pred_len=c["pred_len"],
               rolling_length=c["rolling"], diffusion_schedule="linear", diffusion_steps=c["steps"], beta_start=1e-4, beta_end=0.02, n_z_samples=c["n_z"],
               parallel_sample=c["parallel"], Td_h=c["Td_h"], d_h=c["d_h"], channel_multipliers=c["channel_multipliers"], n_blocks=c["n_blocks"],
               load_pretrain=False, gnn_name="ResGatedGraphConv", gnn_param={"bias": True, "root_weight": True})
    torch.manual_seed(654)
    model = diffusion_models("NsDiff_spatial", net, train_model_select="NsDiff_model").to(device).eval()
    with torch.no_grad():
        for prm in model.parameters():                              # default init leaves the k = T+1 convolutions tiny: spread them
            if prm.dim() == 4 and prm.shape[-1] > 8:
                prm.mul_(3.0)
        model.scaler_mean.fill_(0.0)
        model.scaler_std.fill_(1.0)
    model.refresh_native()
    adj = small_world(c["nodes"], seed=3000)
    src, dst = np.nonzero(adj)
    edge = torch.from_numpy(np.stack([src, dst])).long()
    g = torch.Generator().manual_seed(12)
    W, V = c["n_windows"], c["nodes"]
    hist = torch.cumsum(torch.randn(W, V, c["windows"], 1, generator=g) * 0.1, dim=2).to(device)
    y0 = (torch.randn(W, V, c["pred_len"], 1, generator=g) * 0.3).to(device)
    ctx = model.native_context()

    def step():
        samples = model.sample_windows(hist, edge, y0, seed=7)
        mpv, pm, _ = ctx.reduce(samples)
        return torch.stack([mpv, pm]).cpu()

    dt, curve = _events(step, 2)
    assert bool(torch.isfinite(curve).all())
    # algorithmic FLOPs of one U-Net forward per node row: the DiffSTG count of the same block sequence at time length pred_len (the head adds < 1 %)
    rows = V * c["parallel"]
    fpr = _unet_flops_per_row(model.model.diffussion_model)
    flop_per_window = rows * c["steps"] * fpr
    achieved = flop_per_window * W / dt / 1e12
    rec = {"metric": "rolling windows/sec (N=32, T=50), NsDiff_spatial", "value": W / dt, "unit": "windows/s", "ms_per_step": dt * 1e3, "windows_per_step": W,
           "dtype": "f32 (k = T+1 convolutions, graph-conv Linear maps and TCN blocks as two-term f16 splits on the f16 matrix pipe, fp32 accumulate)",
           "data": "synthetic", "config": {"workload": "NsDiff_spatial on a 200-node small-world graph: windows=100, pred_len=100, d_h=4, multipliers [2,2], 2 blocks "
                                                       "per resolution, Td_h=10, parallel_sample=32 (6400 node rows per window), T=50 steps; f(x) supplied",
                                           "weights": "seeded default init", "noise": "on-device Philox4x32-10"}
           }
    from mpv_amd.bench_common import measured_traffic, multi_kernel_roofline
    tr = measured_traffic("nss")                     # profiles/*/pmc_traffic_nss.json (tools/pmc_pass.sh nss), bytes per window, bound to its binary
    rec["roofline"] = multi_kernel_roofline(
        tr, rec["value"], achieved, "f16x3",
        "the kernels of c5 at time length 100 (stg_tcn_mfma_kernel, nst_gemm_h3_kernel, stg_gate_kernel) + nss_heads / nss_posterior",
        {"flop_per_window": flop_per_window})
    # the same step with f(x) = the native Model_spatial in front of it (round 4; no shipped YAML names its sizes: d_model 128, 4 heads,
    # d_ff 128, 2 + 1 layers, fT_h 10, two graph convolutions over 1280 features)
    try:
        if os.environ.get("MPV_BENCH_NSS_NO_FX"):                      # tools/pmc_pass.sh nss: the counter passes measure the record's own step only
            raise RuntimeError("skipped (MPV_BENCH_NSS_NO_FX)")
        fx = dict(d_model=128, n_heads=4, d_ff=128, e_layers=2, d_layers=1, activation="gelu", p_hidden_dims=[64, 64], p_hidden_layers=2, fT_h=10, spatial_layers=2)
        torch.manual_seed(655)
        full = diffusion_models("NsDiff_spatial", dict(net, **fx), train_model_select="NsDiff_model").to(device).eval()
        with torch.no_grad():
            full.scaler_mean.fill_(0.0)
            full.scaler_std.fill_(1.0)
        full.refresh_native()
        fctx = full.native_context()

        def full_step():
            smp = full.sample_windows(hist, edge, None, seed=7)                      # f(x), g(x), sampler
            mpv, pm, _ = fctx.reduce(smp)
            return torch.stack([mpv, pm]).cpu()

        dtf, fcurve = _events(full_step, 2)
        dfx, _ = _events(lambda: full.spatial_mean_windows(hist, edge), 3)
        rec["with_native_fx"] = {"value": W / dtf, "unit": "windows/s", "ms_per_step": dtf * 1e3, "fx_ms_per_step": dfx * 1e3, "finite": bool(torch.isfinite(fcurve).all()),
                                 "fx": "Model_spatial d_model 128, 4 heads, d_ff 128, 2 enc + 1 dec layers, fT_h 10, 2 x ResGatedGraphConv(1280): parity unpinned"}
        log("other family nsdiff_spatial with native f(x): %.3g windows/s (f(x) %.1f ms per %d windows)" % (W / dtf, dfx * 1e3, W))
        del full
    except Exception as exc:                                                         # noqa: BLE001
        rec["with_native_fx"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    if cpu:
        from oracle import nsdiff_spatial_oracle as O
        from oracle import nsdiff_oracle as NS
        cores = cpu_share()
        torch.set_num_threads(cores)
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        onet = dict(net, device="cpu", parallel_sample=2, n_z_samples=2)
        gz = torch.Generator().manual_seed(3)
        t0 = time.perf_counter()
        outs = O.evaluation_step(sd, onet, hist[0].cpu(), edge, y0[0].cpu(), lambda shape: torch.randn(shape, generator=gz))
        dtc = time.perf_counter() - t0
        rec["cpu_baseline"] = {"value": (2.0 / c["n_z"]) / dtc, "unit": "windows/s", "cores": cores, "kind": "port", "finite": bool(torch.isfinite(outs).all()),
                               "sample": "2 of the 32 graph copies of one window (400 node rows), all T=50 steps, oracle/nsdiff_spatial_oracle.py, {:.1f} s; "
                                         "value = (1/16 window) / time".format(dtc)}
    log("other family nsdiff_spatial: %.3g windows/s" % rec["value"])
    return rec


def _unet_flops_per_row(ugnet):
    td, total, T_cur = int(ugnet.Td_h), 0, ugnet.T
    for st in ugnet.layout():
        m = st["module"]
        if st["kind"] == "res":
            ci, co, T = m.c_in, m.c_out, m.T
            C = td * co
            total += 2 * co * ci * 3 * T + 2 * co * co * 3 * T + 2 * co * td * co * (T + 1) * 2 + 2 * 4 * C * C
            total += 2 * co * ci * T * (2 if ci != co else 0)
        else:
            ch = m.conv.weight.shape[0]
            total += 2 * ch * ch * (3 * T_cur // 2 if st["kind"] == "down" else 4 * T_cur)
            T_cur = (T_cur - 1) // 2 + 1 if st["kind"] == "down" else 2 * T_cur
    total += 2 * ugnet.d_h * 3 * ugnet.F * ugnet.T + 2 * ugnet.d_h * ugnet.d_h * ugnet.T + 2 * ugnet.T * ugnet.T * ugnet.d_h + 4 * ugnet.d_h * ugnet.F * ugnet.T
    return total


def record_tmdm(device, log, cpu=True):
    """The TMDM sampler (pointwise epsilon network without normalisation, CARD posterior) on a C2-shaped batch: 32 windows x 100 nodes x 100
    positions x N = 32, T = 50, prior mean supplied (seeded): the chain kernel of the headline with NORM = false."""
    from mpv_amd import _native
    from mpv_amd.bench_common import cpu_share
    from mpv_amd.models.Diffusion_model.TMDM.tmdm_diffusion_utils import tmdm_coef_rows, tmdm_tables
    c = TMDM
    nf, T = c["nf"], c["steps"]
    g = torch.Generator().manual_seed(0)
    ctx = _native.Context(device)
    lin = lambda o, i: ((torch.rand(o, i, generator=g) * 2 - 1) / i ** 0.5, (torch.rand(o, generator=g) * 2 - 1) / i ** 0.5)
    ws = [lin(128, 2 * nf), lin(128, 128), lin(128, 128)]
    e4 = lin(nf, 128)
    emb = [torch.rand(T + 1, 128, generator=g) for _ in range(3)]
    ctx.load_tmdm(T, nf, [w for w, _ in ws], [b for _, b in ws], emb, e4[0], e4[1], tmdm_coef_rows(tmdm_tables(T, "linear", 1e-4, 0.02)))
    W = c["n_windows"]
    prior = (torch.randn(W, c["nodes"], c["pred_len"], nf, generator=g) * 0.3).to(device)

    def step():
        out = ctx.sample_tmdm(prior, c["n_z"], c["parallel"], seed=5, gemm="f16x3", summary=True, want_samples=False)
        return torch.stack([out[1], out[2]]).cpu()

    ctx.timing_enable(True)
    step()
    ctx.timing()
    dt, curve = _events(step, 3)
    kms, launches = ctx.timing()
    ctx.timing_enable(False)
    assert bool(torch.isfinite(curve).all())
    row_steps = W * c["nodes"] * c["pred_len"] * c["n_z"] * T
    flop = row_steps * 256 * (256 + 3 * nf)                          # 128 (2F) + 2 x 128 x 128 + 128 F multiply-adds per chain-step
    achieved = flop / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
    peak = F16_MATRIX_PEAK_TFLOPS / 3.0
    rec = {"metric": "rolling windows/sec (N=32, T=50), TMDM sampler", "value": W / dt, "unit": "windows/s", "ms_per_step": dt * 1e3, "windows_per_step": W,
           "dtype": "f32 (128x128 layers as 3x f16 MFMA with fp32 accumulate)", "data": "synthetic",
           "config": {"workload": "TMDM epsilon network + CARD posterior on 100 nodes x 100 positions x N=32 samples, T=50 steps, 32 windows per launch; prior mean supplied",
                      "weights": "seeded", "noise": "on-device Philox4x32-10", "summary": "MPV reduction fused into the sampler epilogue"},
           "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": None,
                        "kernel": "nsdiff_chain_f16_kernel<1,false> (NORM = false)", "kernel_avg_ms": kms, "launches": launches, "flop_per_launch": flop,
                        "note": "vector-instruction issue bound like the headline kernel (DESIGN.md 3.1)"}}
    if cpu:
        from oracle import tmdm_oracle as O
        cores = cpu_share()
        torch.set_num_threads(cores)
        sd = {"model.diffussion_model.lin%d.lin.weight" % (i + 1): ws[i][0] for i in range(3)}
        sd.update({"model.diffussion_model.lin%d.lin.bias" % (i + 1): ws[i][1] for i in range(3)})
        sd.update({"model.diffussion_model.lin%d.embed.weight" % (i + 1): emb[i] for i in range(3)})
        sd["model.diffussion_model.lin4.weight"], sd["model.diffussion_model.lin4.bias"] = e4
        try:
            tb = O.schedule_tables(T, "linear", 1e-4, 0.02)
            gz = torch.Generator().manual_seed(4)
            y0 = prior[0].cpu()
            y0_t = y0.repeat_interleave(c["parallel"], dim=0)          # node-major tiling: [B*P, O, F]
            t0 = time.perf_counter()
            outs = O.sample_loop(sd, tb, y0_t, y0_t, T, lambda shape: torch.randn(shape, generator=gz))
            dtc = time.perf_counter() - t0
            rec["cpu_baseline"] = {"value": 1.0 / dtc, "unit": "windows/s", "cores": cores, "kind": "port", "finite": bool(torch.isfinite(outs).all()),
                                   "sample": "one full window (100 nodes x 32 samples x 100 positions, T=50), oracle/tmdm_oracle.py, {:.1f} s".format(dtc)}
        except Exception as exc:                                         # noqa: BLE001
            rec["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    ctx.close()
    log("other family tmdm: %.3g windows/s" % rec["value"])
    return rec
