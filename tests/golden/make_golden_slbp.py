#!/usr/bin/env python3
"""Golden vectors for the SLBP figure-5/6 cache analyses (reference diffusion_model_uncertainy.py:624-1250).

Build container only.  The reference functions are run on seeded synthetic caches written in the reference's own
cache format (no model is loaded: every call below finds its cache on disk), so what is pinned is the cache
bookkeeping and the summaries: which file is read, how elements are classified, the curves and their time points.

    python tests/golden/make_golden_slbp.py
"""
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch
import yaml

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as G                       # noqa: E402


def main():
    ref = G.import_reference()
    H = ref.harness
    g = torch.Generator().manual_seed(2024)
    out = {}
    series = torch.cumsum(torch.randn(30000, 2, generator=g) * 0.05, dim=0) + torch.tensor([3.0, 1.0])
    time = np.arange(30000) * 0.1
    out["series"], out["time"] = G.npy(series), time
    windows, pred_len, sampling_t, step = 40, 20, 10, 7
    inputs, targets, tps = H.build_slbp_sensitivity_windows(series, time, windows, pred_len, sampling_t, step)
    n = len(inputs)
    sampling = [torch.randn(pred_len, 2, 6, generator=g) * (1 + 0.1 * i) for i in range(n)]
    gx = [torch.rand(pred_len, 2, generator=g) * (1 + 0.05 * i) for i in range(n)]
    out["sampling_cache"] = np.stack([G.npy(t) for t in sampling])
    out["gx_cache"] = np.stack([G.npy(t) for t in gx])
    out["dims"] = np.array([windows, pred_len, sampling_t, step, n], dtype=np.int64)

    mpv, dim = H.summarize_slbp_sampling_for_fig6(sampling, pred_dim=1)
    out["fig6_sampling_mpv"], out["fig6_sampling_dim"] = np.array(mpv), np.array(dim, dtype=np.float64)
    out["fig6_gx_mpv"] = np.array(H.summarize_slbp_gx_for_fig6(gx, pred_dim=0))
    rv = H.slbp_raw_window_variance(series, time, windows=windows, sampling_t=sampling_t, sample_window_step=step, pred_dim=1)
    out["raw_var"], out["raw_var_tp"] = np.asarray(rv["variance"]), np.asarray(rv["time_points"])

    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp) / "sweep"
        (root / "models").mkdir(parents=True)
        (root / "datas").mkdir()
        name = "dataset__w40"
        with open(root / "models" / (name + ".yaml"), "w") as f:
            yaml.safe_dump({"dataset": {"windows": windows, "pred_len": pred_len, "sampling_t": sampling_t}}, f)
        # (a) the shared pred_future name holds a gx cache under the LEGACY single-underscore model name
        H._save_tensor_list(gx, H.slbp_fig6_cache_path(root, "dataset_w40", "increase", step, None, "pred_future"))
        r = H.slbp_gx_analysis(root, name, series, time, data_trend="increase", pred_dim=0, sample_window_step=step)
        out["gxa_mpv"], out["gxa_tp"] = np.array(r["gx_mpv"]), np.asarray(r["time_points"])
        out["gxa_cache_name"] = np.array(Path(r["cache_path"]).name)
        # (b) sampling curve: the shared name (current model name) holds gx -> falls through to sampling_pred_future
        H._save_tensor_list(gx, H.slbp_fig6_cache_path(root, name, "increase", step, None, "pred_future"))
        H._save_tensor_list(sampling, H.slbp_fig6_cache_path(root, name, "increase", step, None, "sampling_pred_future"))
        r = H.slbp_sampling_analysis(root, name, series, time, data_trend="increase", pred_dim=1, sample_window_step=step)
        assert r["available"], r["reason"]
        out["sa_mpv"], out["sa_dim"] = np.array(r["mpv"]), np.array(r["intrinsic_dimension"], dtype=np.float64)
        out["sa_tp"], out["sa_cache_name"] = np.asarray(r["time_points"]), np.array(Path(r["cache_path"]).name)
        # (c) figure 5: explicit cache of either kind, window step inferred from the cache length
        for kind, data in (("sampling", sampling[:-3]), ("gx", gx[:-5])):
            cp = root / "datas" / ("fig5_%s.pt" % kind)
            H._save_tensor_list(data, cp)
            r = H.slbp_mpv_analysis(root, name, series, time, cp, pred_dim=0, sample_window_step=3)
            out["fig5_%s_mpv" % kind], out["fig5_%s_tp" % kind] = np.array(r["mpv"]), np.asarray(r["time_points"])
            out["fig5_%s_step" % kind] = np.array(r["sample_window_step"])
            assert r["uncertainty_source"] == kind
    np.savez_compressed(HERE / "slbp_fig_cases.npz", **out)
    f = HERE / "slbp_fig_cases.npz"
    print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    main()
