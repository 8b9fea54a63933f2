#!/usr/bin/env python3
"""End-to-end rate of the reference-shaped entry point `uncertainty_ews` on BASELINE configs[1] (host data file in,
curve + list-of-tensors cache out), i.e. INCLUDING what bench.py leaves out: model load, H2D of the series, D2H of every
sample tensor (232 MB per trajectory) and the torch.save of the cache.  usage: python tools/harness_e2e.py [reps]"""
import os, sys, tempfile, time
from pathlib import Path
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import bench
from helpers import write_model_dir
from mpv_amd import synthetic
from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
C = bench.CFG
dev = torch.device("cuda", 0)
rec = synthetic.biomass_trajectory(C["nodes"], C["ticks"], seed=1000)
full = H.normalize_time_series(rec["ys_dynamic"], dynamic_type="biomass")
sampled, _ = H.sample_time_series(full, rec["ts_dynamic"], C["sampling_t"])
model = bench.build_model(dev, sampled.contiguous())
net = dict(device="cuda", scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=C["rolling"],
           diffusion_schedule="linear", diffusion_steps=C["steps"], beta_start=1e-4, beta_end=0.02, n_z_samples=C["n_z"],
           parallel_sample=C["parallel_sample"], windows=C["windows"], pred_len=C["pred_len"], dataset_nf=C["nf"])
with tempfile.TemporaryDirectory() as tmp:
    tmp = Path(tmp)
    write_model_dir(tmp / "model", {k: v.detach().cpu() for k, v in model.state_dict().items()}, net)
    torch.save({"ys_dynamic": rec["ys_dynamic"], "ts_dynamic": rec["ts_dynamic"], "tp_values": torch.zeros(1)}, tmp / "biomass_run.pt")
    for r in range(reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = H.uncertainty_ews(model_save_file=tmp / "model", data_file=tmp / "biomass_run.pt", dynamic_type="biomass",
                                sampling_t=C["sampling_t"], sample_window_step=C["window_step"], device=dev,
                                force_recompute=True, save_nsdiff_g=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n = len(res["ews"])
        print("%s uncertainty_ews: %d windows in %.3f s = %.1f windows/s (file in, cache file + curve out)" % (
            "warm-up" if r == 0 else "run %d  " % r, n, dt, n / dt), flush=True)
    t0 = time.perf_counter()
    res = H.uncertainty_ews(model_save_file=tmp / "model", data_file=tmp / "biomass_run.pt", dynamic_type="biomass",
                            sampling_t=C["sampling_t"], sample_window_step=C["window_step"], device=dev, save_nsdiff_g=False)
    print("cached second call (summaries from the cache file): %.3f s" % (time.perf_counter() - t0))
