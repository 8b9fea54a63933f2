"""Rolling-window MPV harness on the MI355X-native sampler.

Drop-in for the reference's evaluation_and_analysis/diffusion_model_uncertainy.py (sic): the
public functions keep their names, arguments, error messages, cache-file format (a
`torch.save`d list of CPU tensors per window) and result-dict keys, so the paper's figure
scripts can call `uncertainty_ews(...)` / `slbp_sensitivity_ews(...)` unchanged.

What is different underneath (reference line numbers in each docstring):
  * the reference visits windows one by one (`run_evaluation_cache:323-339`); here all
    windows of a trajectory are gathered from the resident, sub-sampled series on device and
    sampled in ONE chain-kernel launch (windows are independent);
  * scaler transform, sliding-window gather, sigma head, T-step sampling, inverse scaling and
    the variance reduction all run in HIP kernels behind include/mpv_hip.h;
  * with torch.distributed initialised, contiguous blocks of windows are sharded over ranks
    and the per-window scalars are all-gathered (RCCL on GPU ranks, gloo in CPU tests).
"""
from __future__ import annotations

import os
import sys
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np
import torch
import yaml

from .. import distributed as dist_utils

PROJECT_ROOT = Path(os.environ.get("MPV_PROJECT_ROOT", Path.cwd()))

NETWORK_DYNAMICS = {"SIS", "neuronal", "biomass"}
DEFAULT_SAMPLE_WINDOW_STEP = {"SIS": 50, "neuronal": 5, "biomass": 5, "SLBP": 10}
DEFAULT_SAMPLING_T = {"SIS": 0.1, "neuronal": 10, "biomass": 10, "SLBP": 100}
_TICK = 0.1            # spacing of the stored trajectories (reference :120)
# windows per chain launch: bounds the [W, V, O, F, S] sample buffer (and host noise in torch_cpu mode)
MAX_SAMPLE_BYTES = int(os.environ.get("MPV_MAX_SAMPLE_BYTES", 8 << 30))


# ----------------------------------------------------------------------------------------
# paths, configs, caches
# ----------------------------------------------------------------------------------------
def _resolve_project_path(path):
    if path is None:
        return None
    path = Path(path)
    return path if path.is_absolute() else PROJECT_ROOT / path


def _dynamic_name(dynamic_type):
    if dynamic_type is None:
        return None
    text = str(dynamic_type)
    canon = {"sis": "SIS", "slbp": "SLBP", "neuronal": "neuronal", "biomass": "biomass"}
    return canon.get(text.lower(), text)


def _infer_dynamic_type(data_file=None, loaded_data=None):
    """Record keys first, then the file name (reference :69-85)."""
    if loaded_data is not None:
        if "N_values" in loaded_data:
            return "SLBP"
        if "tp_values" in loaded_data:
            return None
    if data_file is None:
        return None
    text = str(data_file).replace("\\", "/").lower()
    for key in ("slbp", "sis", "neuronal", "biomass"):
        if key in text:
            return _dynamic_name(key)
    return None


def normalize_time_series(torch_time_series, dynamic_type=None):
    """-> [Node, T, F].  2-D network records are [T, Node]; other 2-D input is [T, F] (reference :106-116)."""
    data = torch.as_tensor(torch_time_series).float()
    if data.ndim == 3:
        return data
    if data.ndim != 2:
        raise ValueError("time series must have shape [Node, T, F], [T, F], or [T, Node].")
    if _dynamic_name(dynamic_type) in NETWORK_DYNAMICS:
        return data.t().unsqueeze(-1)
    return data.unsqueeze(0)


def load_dynamic_data(data_file, dynamic_type=None, map_location="cpu"):
    """Read a `{ys_dynamic, ts_dynamic, ...}` record (reference :88-103)."""
    loaded = torch.load(data_file, map_location=map_location)
    dynamic_type = _dynamic_name(dynamic_type) or _infer_dynamic_type(data_file=data_file, loaded_data=loaded)
    if "ys_dynamic" not in loaded or "ts_dynamic" not in loaded:
        raise KeyError("data_file must contain 'ys_dynamic' and 'ts_dynamic'.")
    return {
        "torch_time_series": normalize_time_series(loaded["ys_dynamic"], dynamic_type=dynamic_type),
        "time_data": loaded["ts_dynamic"],
        "dynamic_type": dynamic_type,
        "loaded_data": loaded,
    }


def read_model_config(model_save_file):
    config_path = Path(model_save_file) / "model_trained.yaml"
    if not config_path.exists():
        raise FileNotFoundError("model config not found: {}".format(config_path))
    with open(config_path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f)


def _train_select(method_config):
    train = method_config.get("train") if method_config else None
    return train.get("train_model_select") if train is not None else None


def _default_device(device):
    return device or torch.device("cuda" if torch.cuda.is_available() else "cpu")


def load_model_from_dir(model_save_file, device=None, infer_params=None, method_config=None):
    """<dir>/model_trained{,.yaml} -> (model.eval(), merged net_param) (reference :193-215)."""
    from ..utils.utils import load_diffusion_model

    model_save_file = Path(model_save_file)
    method_config = method_config or read_model_config(model_save_file)
    model, net_param = load_diffusion_model(str(model_save_file / "model_trained"), device=_default_device(device),
                                            infer_para=infer_params, train_model_select=_train_select(method_config))
    return model.eval(), net_param


def load_sensitivity_model(model_root, model_name, device=None, infer_params=None):
    """<root>/models/<name>{,.yaml} layout of the SLBP sensitivity sweeps (reference :425-455)."""
    from ..utils.utils import load_diffusion_model

    model_root = _resolve_project_path(model_root)
    config_path = model_root / "models" / "{}.yaml".format(model_name)
    model_path = model_root / "models" / model_name
    if not config_path.exists():
        raise FileNotFoundError("model config not found: {}".format(config_path))
    if not model_path.exists():
        raise FileNotFoundError("model checkpoint not found: {}".format(model_path))
    with open(config_path, "r", encoding="utf-8") as f:
        method_config = yaml.safe_load(f)
    model, net_param = load_diffusion_model(str(model_path), device=_default_device(device), infer_para=infer_params,
                                            train_model_select=_train_select(method_config))
    return model.eval(), method_config, net_param


def default_cache_dir(model_save_file, dynamic_type):
    if model_save_file is not None:
        return Path(model_save_file)
    return PROJECT_ROOT / "ews_results" / "model_uncertainy_cache" / "model" / (_dynamic_name(dynamic_type) or "unknown")


def data_cache_name(data_file, suffix=""):
    if data_file is None:
        return "data{}.pt".format(suffix)
    data_path = Path(data_file)
    return "{}{}{}".format(data_path.stem, suffix, data_path.suffix or ".pt")


def resolve_cache_path(cache_path, model_save_file, data_file, dynamic_type, suffix=""):
    if cache_path is None:
        return default_cache_dir(model_save_file, dynamic_type) / data_cache_name(data_file, suffix=suffix)
    cache_path = _resolve_project_path(cache_path)
    return cache_path if cache_path.suffix == ".pt" else cache_path / data_cache_name(data_file, suffix=suffix)


def resolve_figure_path(cache_file_path):
    return Path(cache_file_path).with_suffix(".png")


def slbp_sensitivity_cache_path(model_root, model_name, data_trend, sample_window_step=10):
    return _resolve_project_path(model_root) / "datas" / "{}_pred_future_{}_{}.pt".format(
        model_name, data_trend, sample_window_step)


def _save_tensor_list(data_list, cache_path):
    """The reference's cache format: one torch.save of a Python list of CPU tensors (:252-256)."""
    cache_path = Path(cache_path)
    cache_path.parent.mkdir(parents=True, exist_ok=True)
    with open(cache_path, "wb") as f:
        torch.save([t.detach().cpu() for t in data_list], f)


def _load_tensor_list(cache_path):
    with open(cache_path, "rb") as f:
        data_list = torch.load(f, map_location="cpu")
    if not isinstance(data_list, list):
        raise TypeError("cache file must contain a list of tensors: {}".format(cache_path))
    return data_list


def _save_on_primary(data_list, cache_path):
    """One writer per cache file: rank 0 saves, the others wait until the file is complete."""
    if dist_utils.is_primary():
        _save_tensor_list(data_list, cache_path)
    dist_utils.barrier()


def read_sensitivity_pred_future_cache(cache_path):
    try:
        return _load_tensor_list(cache_path)
    except Exception as exc:          # a corrupt cache is recomputed, as in the reference (:494-499)
        print("warning: failed to read cache {}, recomputing ({})".format(cache_path, exc))
        return None


# ----------------------------------------------------------------------------------------
# sub-sampling and window arithmetic (integer-exact)
# ----------------------------------------------------------------------------------------
def sampling_interval_from_t(sampling_t):
    """stride = 1 if sampling_t <= 0.1 else max(1, int(sampling_t / 0.1)) -- float division then
    truncation, reproduced literally (10/0.1 -> 100, 0.3/0.1 -> 2) (reference :119-125)."""
    if sampling_t is None or sampling_t <= _TICK:
        return 1
    return max(1, int(sampling_t / _TICK))


def sample_time_series(torch_time_series, time_data, sampling_t, return_numpy_time=True):
    stride = sampling_interval_from_t(sampling_t)
    sampled_time = torch.as_tensor(time_data)[::stride]
    if return_numpy_time:
        sampled_time = sampled_time.cpu().detach().numpy()
    return torch_time_series[:, ::stride, :], sampled_time


def torch_data_preprocessing_like_slbp(time_data, sampling_t, return_numpy=False):
    sampled = torch.as_tensor(time_data)[:: sampling_interval_from_t(sampling_t)]
    return sampled.cpu().detach().numpy() if return_numpy else sampled


def sliding_window_count(sampled_length, windows, sample_window_step):
    return 0 if sampled_length < windows else (sampled_length - windows) // sample_window_step + 1


def build_sliding_windows(torch_time_series, time_data, windows, sample_window_step):
    """[V, L, F] -> (tuple of W views [V, windows, F], time stamp of each window's last sample)
    (reference :137-147).  Window w = samples [w*step, w*step + windows)."""
    if torch_time_series.ndim != 3:
        raise ValueError("torch_time_series must have shape [Node_num, T_obs_num, F].")
    length = torch_time_series.shape[1]
    if length < windows:
        raise ValueError("T_obs_num ({}) is shorter than windows ({}).".format(length, windows))
    count = sliding_window_count(length, windows, sample_window_step)
    views = tuple(torch_time_series[:, w * sample_window_step: w * sample_window_step + windows, :] for w in range(count))
    return views, np.asarray(time_data)[windows - 1:: sample_window_step]


def build_slbp_sensitivity_windows(torch_time_series, time_data, windows, pred_len, sampling_t, sample_window_step):
    """Single series [T, F]: input windows and the matching future windows (reference :467-483)."""
    series = torch_data_preprocessing_like_slbp(torch_time_series, sampling_t=sampling_t)
    sampled_time = torch_data_preprocessing_like_slbp(time_data, sampling_t=sampling_t, return_numpy=True)
    time_points = sampled_time[windows - 1:: sample_window_step]
    length = series.shape[0]
    if length < windows:      # torch.unfold's error in the reference
        raise RuntimeError("maximum size for tensor at dimension 0 is {} but size is {}".format(length, windows))
    n_in = sliding_window_count(length, windows, sample_window_step)
    input_datas = tuple(series[w * sample_window_step: w * sample_window_step + windows, :] for w in range(n_in))
    future = series[windows:, :]
    if future.shape[0] >= pred_len:
        n_out = sliding_window_count(future.shape[0], pred_len, sample_window_step)
        pred_datas = tuple(future[w * sample_window_step: w * sample_window_step + pred_len, :] for w in range(n_out))
    else:
        pred_datas = ()
    return input_datas, pred_datas, time_points


def default_sample_window_step(dynamic_type, task_model=None, dataset_config=None):
    dataset_config = dataset_config or {}
    if task_model == "DiffSTG" and dataset_config.get("interval_step") is not None:
        return dataset_config["interval_step"]
    return DEFAULT_SAMPLE_WINDOW_STEP.get(dynamic_type, 10)


def infer_sample_window_step_from_cache(sampled_length, windows, cache_len, fallback_step):
    """Recover the window step a cache was written with from its length (reference :163-182)."""
    if cache_len <= 0 or sampled_length < windows:
        return fallback_step
    if cache_len == 1 or sliding_window_count(sampled_length, windows, fallback_step) == cache_len:
        return fallback_step
    span = sampled_length - windows
    low = int(np.floor(span / cache_len)) + 1
    high = int(np.floor(span / (cache_len - 1)))
    fits = [s for s in range(max(1, low), max(1, high) + 1) if sliding_window_count(sampled_length, windows, s) == cache_len]
    return min(fits, key=lambda s: (abs(s - fallback_step), -s)) if fits else fallback_step


# ----------------------------------------------------------------------------------------
# device execution
# ----------------------------------------------------------------------------------------
def _uses_scaler(model) -> bool:
    # the YAML string 'None' is truthy in the reference and so is here
    return getattr(model, "scaler", None) is not None


def _windows_per_launch(model, nodes: int) -> int:
    per_window = 4 * nodes * model.pred_len * model.dataset_nf * int(model.configs.n_z_samples)
    if getattr(model, "noise_source", "philox") == "torch_cpu":
        per_window *= 1 + model.diffusion_steps
    return max(1, MAX_SAMPLE_BYTES // max(per_window, 1))


def _stack_windows(window_list: Sequence[torch.Tensor]) -> torch.Tensor:
    """W tensors [V, windows, F] -> one [W*V, windows, F] batch (windows become extra rows: the
    sampler treats every row independently, exactly as the reference treats nodes)."""
    return torch.stack([torch.as_tensor(w).float() for w in window_list], dim=0).flatten(0, 1)


# Test hook: callable(model, windows, pred_len, scale) -> (samples [w, V, pred_len, F, S], mpv [w], pred_mean [w]) that
# replaces the device computation of ONE shard (tests/test_distributed_gloo.py injects the CPU oracle here so that the
# sharding, gathering and cache logic around it run under gloo without a GPU).  Never set by the product.
SHARD_COMPUTE = None
# The same for the sample-axis split: callable(model, windows, pred_len, scale, s_lo, s_hi) -> samples [W, V, pred_len, F, s_hi - s_lo].
SAMPLE_SHARD_COMPUTE = None


def _skip_host_noise(model, n_windows: int, nodes: int):
    """torch_cpu noise under sharding: the reference draws window after window from ONE stream, so a rank whose block
    starts at window `n_windows` first consumes the draws of the windows before it (drawn and dropped one window at a
    time: the CPU generator has no skip-ahead)."""
    if getattr(model, "noise_source", "philox") != "torch_cpu":
        return
    for _ in range(n_windows):
        model.host_noise(1, nodes)


# sequences (DiffusionTS) / graph-node rows (DiffSTG) per library call of the batched-history model families
MAX_BATCH_ROWS = int(os.environ.get("MPV_MAX_BATCH_ROWS", 4096))


def _native_shard_batched(model, windows, pred_len, scale, graph=None):
    """DiffusionTS / DiffSTG: their `sample_windows` takes [W, V, history, F] and runs W evaluation_steps in one call."""
    V = int(windows[0].shape[0])
    ctx = model.native_context()
    inv_std = model.scaler_std.to(ctx.device).float().contiguous() if _uses_scaler(model) else None
    inv_mean = model.scaler_mean.to(ctx.device).float().contiguous() if _uses_scaler(model) else None
    is_nss = graph is not None and _is_spatial(model)           # NsDiff_spatial: windows of one graph, f(x) = the native Model_spatial
    is_stg = graph is not None and not is_nss
    hist_len = model.T_h if is_stg else model.windows
    if is_nss:
        per_window = V * int(model.configs.parallel_sample)
    else:
        per_window = V * (int(model.parallel_sampling) if is_stg else min(int(model.parallel_sample), int(model.n_z_samples)))
    chunk = max(1, MAX_BATCH_ROWS // max(per_window, 1))
    preds, mpv_parts, mean_parts = [], [], []
    for start in range(0, len(windows), chunk):
        part = windows[start: start + chunk]
        hist = torch.stack([torch.as_tensor(w).float()[:, :hist_len, :] for w in part]).to(ctx.device)         # [w, V, hist, F]
        if scale:
            hist = (hist - model.scaler_mean.to(ctx.device)) / model.scaler_std.to(ctx.device)
        samples = model.sample_windows(hist, graph.edge_index) if graph is not None else model.sample_windows(hist)
        samples = samples[:, :, -pred_len:].contiguous()                                                       # [w, V, pred_len, F, S]
        mpv, pm, _ = ctx.reduce(samples, inv_std, inv_mean)
        mpv_parts.append(mpv)
        mean_parts.append(pm)
        preds.append(samples if dist_utils.active() else samples.cpu())
    return torch.cat(preds, dim=0), torch.cat(mpv_parts), torch.cat(mean_parts)


def _is_spatial(model) -> bool:
    """NsDiff_model_spatial: samples over a graph with the NsDiff posterior (its `sample_windows(history, edge_index)` runs f(x), g(x) and the sampler)."""
    return hasattr(model, "spatial_mean_windows")


def _native_shard(model, windows, pred_len, scale, keep_samples=True, graph=None):
    """This rank's windows on the device: as few launches as memory allows.  keep_samples=False (curve-only runs of a
    stacked model whose horizon is not cut) never materialises the sample tensor: the summary comes out of the sampler."""
    if getattr(model, "batched_history", False):
        return _native_shard_batched(model, windows, pred_len, scale, graph=graph)
    V = int(windows[0].shape[0])
    ctx = model.native_context()
    inv_std = model.scaler_std.to(ctx.device).float().contiguous() if _uses_scaler(model) else None
    inv_mean = model.scaler_mean.to(ctx.device).float().contiguous() if _uses_scaler(model) else None
    preds, mpv_parts, mean_parts = [], [], []
    if not hasattr(model, "_prior_variance"):
        # TMDM: the prior network (native ns-Transformer VAE mean, or a module the caller assigned) over all rows of a block of windows in
        # one call, then ONE sampler launch for the block (the reference loops window by window, tmdm_adapter.py:116-155)
        block = max(1, 4096 // max(V, 1))
        for start in range(0, len(windows), block):
            part = windows[start: start + block]
            rows = _stack_windows(part).to(ctx.device)                        # [w*V, windows, F]
            if scale:
                rows = model.scaler_transform(rows)
            if getattr(model, "cond_pred_model", None) is None:
                raise NotImplementedError("TMDM without a prior network: assign one to `cond_pred_model`")
            with torch.no_grad():
                _, y0, _, _ = model.cond_pred_model(rows, None, model._decoder_input(rows), None)
            prior = y0[:, -model.pred_len:, :].reshape(len(part), V, model.pred_len, model.dataset_nf)
            samples = model.sample_windows(prior)[:, :, -pred_len:].contiguous()          # [w, V, pred_len, F, S]
            mpv, pm, _ = ctx.reduce(samples, inv_std, inv_mean)
            mpv_parts.append(mpv)
            mean_parts.append(pm)
            preds.append(samples if dist_utils.active() else samples.cpu())
    else:
        chunk = _windows_per_launch(model, V)
        for start in range(0, len(windows), chunk):
            part = windows[start: start + chunk]
            rows = _stack_windows(part).to(ctx.device)                        # [w*V, windows, F]
            noise = _host_noise_for(model, len(part), V)
            if not keep_samples and pred_len == model.pred_len:
                # windows as the launch's window axis so that the fused summary is per window
                rows4 = rows.reshape(len(part), V, rows.shape[1], rows.shape[2])
                _, mpv, pm = _summary_only(model, rows4, scale, noise)
                mpv_parts.append(mpv)
                mean_parts.append(pm)
                preds.append(torch.zeros((len(part), 0), device=ctx.device))
                continue
            # one pseudo-window over w*V rows; the scaler is fused into the sigma-head load
            samples, _ = model.sample_windows(rows, 1, 0, 1, scale_input=scale, noise=noise)
            samples = samples.reshape(len(part), V, model.pred_len, model.dataset_nf, -1)[:, :, -pred_len:]
            samples = samples.contiguous()
            mpv, pm, _ = ctx.reduce(samples, inv_std, inv_mean)
            mpv_parts.append(mpv)
            mean_parts.append(pm)
            preds.append(samples if dist_utils.active() else samples.cpu())     # sharded: gathered from device memory
    return torch.cat(preds, dim=0), torch.cat(mpv_parts), torch.cat(mean_parts)


def _native_sample_shard(model, windows, pred_len, scale, s_lo, s_hi, seed):
    """Sample-axis split (fewer windows than ranks): ALL windows, this rank's samples [s_lo, s_hi) of each.  Stacked NsDiff family only.
    torch_cpu noise: every rank draws the reference's full stream and keeps the draws of its own samples (whole chunks, or a slice
    of the parallel_sample axis of one chunk), so the union over ranks is bit-for-bit the single-process result; Philox: one shared
    seed, rank-distinct keys."""
    cfg = model.configs
    S, P = model._check_sampling_config()
    per = s_hi - s_lo
    V = int(windows[0].shape[0])
    ctx = model.native_context()
    rows = _stack_windows(windows).to(ctx.device)                             # [W*V, windows, F]
    noise = None
    if getattr(model, "noise_source", "philox") == "torch_cpu":
        z = model.host_noise(len(windows), V)                                 # [W, C, T, V*P, O, F]
        W_, C_, T_, _, O_, F_ = z.shape
        z = z.reshape(W_, C_, T_, V, P, O_, F_)
        if per % P == 0:
            z = z[:, s_lo // P: s_hi // P]
            p_loc = P
        else:
            z = z[:, s_lo // P: s_lo // P + 1, :, :, s_lo % P: s_lo % P + per]
            p_loc = per
        c_loc = z.shape[1]
        # stacked launch over W*V rows: [1, C', T, (W*V)*P', O, F]
        noise = z.permute(1, 2, 0, 3, 4, 5, 6).reshape(1, c_loc, T_, W_ * V * p_loc, O_, F_).contiguous()
    else:
        p_loc = P if per % P == 0 else per
    keep = (cfg.n_z_samples, cfg.parallel_sample)
    cfg.n_z_samples, cfg.parallel_sample = per, p_loc
    try:
        samples, _ = model.sample_windows(rows, 1, 0, 1, scale_input=scale, noise=noise, seed=None if noise is not None else seed + dist_utils.rank())
    finally:
        cfg.n_z_samples, cfg.parallel_sample = keep
    return samples.reshape(len(windows), V, model.pred_len, model.dataset_nf, per)[:, :, -pred_len:].contiguous()


def _sample_axis(model, graph=None):
    """(n_samples, chunk, whole_chunks_only) of a model family's sample axis: NsDiff and TMDM sample pointwise in chunks of
    `parallel_sample` (a chunk may be cut); a DiffusionTS chunk is one batch with shared normalisers and a DiffSTG chunk one sequential pass
    of `parallel_sampling` graph copies (whole chunks only)."""
    if graph is not None and _is_spatial(model):             # NsDiff_spatial: a chunk = one pass of parallel_sample graph copies (whole chunks only)
        return int(model.configs.n_z_samples), int(model.configs.parallel_sample), True
    if graph is not None:
        return int(model.parallel_sampling) * int(model.sequential_sampling), int(model.parallel_sampling), True
    if hasattr(model, "_prior_variance") and hasattr(model, "configs"):
        return int(model.configs.n_z_samples), int(model.configs.parallel_sample), False
    if hasattr(model, "_chunking"):
        S, P = model._chunking()
        return S, P, bool(getattr(model, "batched_history", False))
    return None


def _native_sample_shard_other(model, windows, pred_len, scale, s_lo, s_hi, seed, graph=None):
    """Sample-axis shard of the DiffusionTS / DiffSTG / TMDM families: ALL windows, this rank's samples [s_lo, s_hi) -- whole chunks
    (TMDM: also a slice of one chunk's parallel_sample axis).  torch_cpu noise: every rank draws the reference's full stream and keeps the
    draws of its own chunks, so the union over ranks is the single-process result; Philox: the shared seed + the rank."""
    S, P, _ = _sample_axis(model, graph)
    per = s_hi - s_lo
    V = int(windows[0].shape[0])
    ctx = model.native_context()
    torch_cpu = getattr(model, "noise_source", "philox") == "torch_cpu"
    is_nss = graph is not None and _is_spatial(model)
    is_stg, is_dts = graph is not None and not is_nss, graph is None and getattr(model, "batched_history", False)
    if is_nss:
        # NsDiff_spatial: f(x) and g(x) for all windows on every rank (cheap beside the sampler), this rank's chunks of the chains
        hist = torch.stack([torch.as_tensor(w).float()[:, :model.windows, :] for w in windows]).to(ctx.device)      # [W, V, windows, F]
        if scale:
            hist = (hist - model.scaler_mean.to(ctx.device)) / model.scaler_std.to(ctx.device)
        noise = model.host_noise(len(windows), V)[:, s_lo // P: s_hi // P].contiguous() if torch_cpu else None     # [W, C, T, V*P, O, F]: chunk axis second
        keep = model.configs.n_z_samples
        model.configs.n_z_samples = per
        try:
            samples = model.sample_windows(hist, graph.edge_index, None, noise=noise, seed=None if torch_cpu else seed + dist_utils.rank())
        finally:
            model.configs.n_z_samples = keep
        return samples[:, :, -pred_len:].contiguous()
    if is_stg or is_dts:
        hist_len = model.T_h if is_stg else model.windows
        hist = torch.stack([torch.as_tensor(w).float()[:, :hist_len, :] for w in windows]).to(ctx.device)          # [W, V, hist, F]
        if scale:
            hist = (hist - model.scaler_mean.to(ctx.device)) / model.scaler_std.to(ctx.device)
        noise = model.host_noise(len(windows), V)[s_lo // P: s_hi // P].contiguous() if torch_cpu else None      # chunk axis first in both layouts
        if is_stg:
            keep = model.sequential_sampling
            model.sequential_sampling = per // P
            try:
                samples = model.sample_windows(hist, graph.edge_index, noise=noise, seed=None if torch_cpu else seed + dist_utils.rank())
            finally:
                model.sequential_sampling = keep
        else:
            keep = model.n_z_samples
            model.n_z_samples = per
            try:
                samples = model.sample_windows(hist, noise=noise, seed=None if torch_cpu else seed + dist_utils.rank())
            finally:
                model.n_z_samples = keep
        return samples[:, :, -pred_len:].contiguous()
    # TMDM: the prior over all rows once, then this rank's share of the chains
    rows = _stack_windows(windows).to(ctx.device)
    if scale:
        rows = model.scaler_transform(rows)
    if getattr(model, "cond_pred_model", None) is None:
        raise NotImplementedError("TMDM without a prior network: assign one to `cond_pred_model`")
    with torch.no_grad():
        _, y0, _, _ = model.cond_pred_model(rows, None, model._decoder_input(rows), None)
    prior = y0[:, -model.pred_len:, :].reshape(len(windows), V, model.pred_len, model.dataset_nf)
    noise, p_loc = None, (P if per % P == 0 else per)
    if torch_cpu:
        z = model.host_noise(len(windows), V)                                 # [W, C, T, V*P, O, F]
        W_, C_, T_, _, O_, F_ = z.shape
        z = z.reshape(W_, C_, T_, V, P, O_, F_)
        z = z[:, s_lo // P: s_hi // P] if per % P == 0 else z[:, s_lo // P: s_lo // P + 1, :, :, s_lo % P: s_lo % P + per]
        noise = z.reshape(W_, z.shape[1], T_, V * p_loc, O_, F_).contiguous()
    keep = (model.n_z_samples, model.parallel_sample)
    model.n_z_samples, model.parallel_sample = per, p_loc
    try:
        samples = model.sample_windows(prior, noise=noise, seed=None if torch_cpu else seed + dist_utils.rank())
    finally:
        model.n_z_samples, model.parallel_sample = keep
    return samples[:, :, -pred_len:].contiguous()


def _sample_split(model, window_list, pred_len, scale, keep_samples, graph=None):
    """More ranks than windows: shard the SAMPLE axis.  Every rank samples all windows with n_z_samples / world samples; per-row
    (mean, M2) of the inverse-scaled samples are merged with `distributed.merge_row_moments` (one all-gather, Chan's update in rank
    order: every rank gets the same curve), and when the caller keeps the samples they are all-gathered along the sample axis."""
    S = _sample_axis(model, graph)[0]
    s_lo, s_hi = dist_utils.sample_bounds(S)
    seed = 0
    if getattr(model, "noise_source", "philox") != "torch_cpu":              # (torch_cpu: the shared CPU stream must stay untouched)
        seed = dist_utils.broadcast_int(int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()))
    stacked = graph is None and hasattr(model, "_prior_variance") and hasattr(model, "configs")
    if SAMPLE_SHARD_COMPUTE is not None:
        local = SAMPLE_SHARD_COMPUTE(model, list(window_list), pred_len, scale, s_lo, s_hi)
    elif stacked:
        local = _native_sample_shard(model, list(window_list), pred_len, scale, s_lo, s_hi, seed)
    else:
        local = _native_sample_shard_other(model, list(window_list), pred_len, scale, s_lo, s_hi, seed, graph=graph)
    x = local.double()
    if _uses_scaler(model):                                                   # summarize_pred_future_list inverse-scales first (:286-303)
        x = x * model.scaler_std.to(x.device).double().view(1, 1, 1, -1, 1) + model.scaler_mean.to(x.device).double().view(1, 1, 1, -1, 1)
    mean = x.mean(dim=-1)
    m2 = ((x - mean.unsqueeze(-1)) ** 2).sum(dim=-1)
    var, mean = dist_utils.merge_row_moments(mean, m2, s_hi - s_lo)            # [W, V, O, F] each
    mpv = var.flatten(1).mean(dim=1).float().cpu().numpy()
    pm = mean.flatten(1).mean(dim=1).float().cpu().numpy()
    if not keep_samples:
        return None, mpv, pm
    return list(dist_utils.gather_samples(local).unbind(0)), mpv, pm


def _summary_only(model, rows4, scale, noise):
    """[w, V, windows, F] stacked windows -> (None, mpv [w], pred_mean [w]) without a sample tensor: sigma head over
    w*V rows, then ONE sampler launch with w windows of V nodes whose epilogue reduces every (node, horizon, feature) row."""
    w, V = rows4.shape[0], rows4.shape[1]
    ctx = model.native_context()
    flat = rows4.reshape(w * V, rows4.shape[2], rows4.shape[3]).contiguous()
    gx = model._prior_variance(ctx, flat, 1, 0, 1, scale).reshape(w, V, model.pred_len, model.dataset_nf)
    y0hat = None
    if getattr(model, "needs_backbone", False):
        wins = (flat - model.scaler_mean.to(flat.device)) / model.scaler_std.to(flat.device) if scale else flat
        y0hat = model._backbone_mean(wins).to(ctx.device).float().reshape(gx.shape).contiguous()
    S, P = model._check_sampling_config()
    std = mean = None
    if _uses_scaler(model):
        std, mean = model.scaler_std.to(ctx.device).float().contiguous(), model.scaler_mean.to(ctx.device).float().contiguous()
    if noise is not None:                                    # [1, C, T, (w*V)*P, O, F] of the stacked launch -> [w, C, T, V*P, O, F]
        _, C_, T_, _, O_, F_ = noise.shape
        noise = noise.reshape(C_, T_, w, V * P, O_, F_).permute(2, 0, 1, 3, 4, 5).contiguous().to(ctx.device)
        seed = 0
    else:
        seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    _, mpv, pm, _ = ctx.sample(gx.contiguous(), y0hat, S, P, noise=noise, seed=seed, gemm=model.gemm_mode, summary=True,
                               want_samples=False, std=std, mean=mean)
    return None, mpv, pm


def _sample_stacked(model, window_list, pred_len, scale: bool, keep_samples: bool = True, graph=None):
    """All windows of a trajectory -> (list of W CPU tensors [V, pred_len, F, S], MPV [W], pred_mean [W]).

    With torch.distributed active every rank computes a contiguous block of windows (`distributed.shard_bounds`), then
    the per-window scalars AND the sample tensors are all-gathered, so every rank returns the complete, identical
    result: the reference's list holds one entry per window (reference :323-339) and the cache written from it must too.
    With fewer windows than ranks (a single SLBP window, a short real-data series) the sample axis is split instead
    (`_sample_split`)."""
    n = len(window_list)
    if n == 0:
        return [], np.zeros(0, np.float32), np.zeros(0, np.float32)
    axis = _sample_axis(model, graph) if dist_utils.active() else None
    if axis is not None and dist_utils.sample_split(n, axis[0], axis[1], whole_chunks_only=axis[2]):
        return _sample_split(model, window_list, pred_len, scale, keep_samples or getattr(model, "batched_history", False), graph=graph)
    lo, hi = dist_utils.shard_bounds(n)
    V = int(window_list[0].shape[0])
    compute = SHARD_COMPUTE or _native_shard
    if hi > lo:
        if lo > 0:
            _skip_host_noise(model, lo, V)
        if graph is not None and SHARD_COMPUTE is None:
            samples, mpv_local, mean_local = compute(model, list(window_list[lo:hi]), pred_len, scale, graph=graph)
        elif keep_samples or getattr(model, "batched_history", False):
            samples, mpv_local, mean_local = compute(model, list(window_list[lo:hi]), pred_len, scale)
        else:
            samples, mpv_local, mean_local = compute(model, list(window_list[lo:hi]), pred_len, scale, keep_samples=False)
    else:                                   # more ranks than windows: an empty block still joins the collectives
        cfg = getattr(model, "configs", model)
        S = int(getattr(cfg, "n_z_samples", getattr(model, "n_z_samples", 1)))
        if graph is not None and not _is_spatial(model):
            S = int(model.parallel_sampling) * int(model.sequential_sampling)
        nccl = dist_utils.active() and torch.distributed.get_backend() == "nccl"
        dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
        samples = torch.zeros((0, V, pred_len, int(window_list[0].shape[-1]), S), device=dev)
        mpv_local = mean_local = torch.zeros(0, device=dev)
    mpv_all, mean_all = dist_utils.gather_curve(n, lo, hi, mpv_local, mean_local)
    if not keep_samples and not getattr(model, "batched_history", False):
        return None, mpv_all, mean_all
    full = dist_utils.gather_windows(n, lo, hi, samples)
    return list(full.unbind(0)), mpv_all, mean_all


def _host_noise_for(model, n_windows: int, nodes: int):
    """torch_cpu mode: the reference draws window by window, so draw [n_windows] evaluation_step
    worth of normals and re-tile them for the stacked (1 window x n_windows*nodes rows) launch."""
    if getattr(model, "noise_source", "philox") != "torch_cpu":
        return None
    z = model.host_noise(n_windows, nodes)               # [W, C, T, V*P, O, F]
    W, C, T, VP, O, F = z.shape
    P = VP // nodes
    # stacked launch: rows = (w, v) pairs, so its draw tensor is [1, C, T, (W*V)*P, O, F]
    z = z.reshape(W, C, T, nodes, P, O, F).permute(1, 2, 0, 3, 4, 5, 6).reshape(1, C, T, W * nodes * P, O, F)
    return z.contiguous()


class _DeviceCurve(list):
    """pred_future_list whose MPV / pred_mean were already reduced on device (same list API)."""
    ews: Optional[np.ndarray] = None
    pred_mean: Optional[np.ndarray] = None


def run_evaluation_cache(model, timeseries_datas, pred_len, cache_path, device, force_recompute=False, max_windows=None,
                         store_samples=True):
    """Windows -> list of CPU tensors [V, pred_len, F, n_z_samples], cached (reference :323-339).

    The reference leaves `time_series_trans` unbound when `model.scaler is None`; here an
    unscaled model simply samples the raw window.  store_samples=False (extension): curve only -- nothing is cached, the
    returned list is empty and carries the device-reduced `ews` / `pred_mean`; no sample tensor is ever written."""
    cache_path = Path(cache_path)
    if cache_path.exists() and not force_recompute:
        return _load_tensor_list(cache_path)
    windows = list(timeseries_datas[:max_windows] if max_windows is not None else timeseries_datas)
    if not store_samples:
        with torch.no_grad():
            _, mpv, pm = _sample_stacked(model, windows, pred_len, scale=_uses_scaler(model), keep_samples=False)
        out = _DeviceCurve()
        out.ews, out.pred_mean = mpv, pm
        return out
    with torch.no_grad():
        preds, mpv, pm = _sample_stacked(model, windows, pred_len, scale=_uses_scaler(model))
    out = _DeviceCurve(preds)
    out.ews, out.pred_mean = mpv, pm
    _save_on_primary(list(out), cache_path)
    return out


def run_slbp_sensitivity_cache(model, input_datas, cache_path, device, force_recompute=False, max_windows=None):
    """Single-series windows [windows, F] -> list of [pred_len, F, n_z_samples] (reference :502-526)."""
    cache_path = Path(cache_path)
    if cache_path.exists() and not force_recompute:
        cached = read_sensitivity_pred_future_cache(cache_path)
        if cached is not None:
            return cached
    windows = [torch.as_tensor(w).float().unsqueeze(0) for w in
               (input_datas[:max_windows] if max_windows is not None else input_datas)]
    with torch.no_grad():
        preds, _, _ = _sample_stacked(model, windows, model.pred_len, scale=_uses_scaler(model))
    preds = [p.squeeze(0) for p in preds]
    _save_on_primary(preds, cache_path)
    return preds


class _Graph:
    """edge_index [2, E] (both directions of every undirected edge, source row 0 / target row 1) + num_nodes: what the reference gets
    from torch_geometric.utils.from_networkx (reference :342-351), without torch_geometric."""
    def __init__(self, edge_index, num_nodes):
        self.edge_index, self.num_nodes, self.x = edge_index, int(num_nodes), None

    def clone(self):
        return _Graph(self.edge_index.clone(), self.num_nodes)


def load_diffstg_graph(graph_file):
    import networkx as nx
    if graph_file is None:
        raise ValueError("graph_file is required for DiffSTG.")
    graph_file = _resolve_project_path(graph_file)
    nx_g = nx.convert_node_labels_to_integers(nx.read_graphml(graph_file))
    directed = nx_g if nx_g.is_directed() else nx_g.to_directed()
    edges = list(directed.edges)
    edge_index = torch.tensor(edges, dtype=torch.long).t().contiguous() if edges else torch.zeros((2, 0), dtype=torch.long)
    return _Graph(edge_index, nx_g.number_of_nodes())


def normalize_diffstg_pred_future_list(pred_future_list):
    out = []
    for pred_future in pred_future_list:
        pred_future = torch.as_tensor(pred_future).detach().cpu()
        if pred_future.ndim == 3:
            pred_future = pred_future.unsqueeze(-2)
        if pred_future.ndim != 4:
            raise ValueError("DiffSTG pred_future must have shape [Node, pred_len, F, samples] "
                             "or legacy [Node, pred_len, samples], got {}".format(tuple(pred_future.shape)))
        out.append(pred_future)
    return out


def run_diffstg_evaluation_cache(model, timeseries_datas, pred_len, graph_data, cache_path, device, force_recompute=False, max_windows=None):
    """Windows -> list of CPU tensors [V, pred_len, 1, parallel*sequential], cached (reference :369-397): all windows as disjoint
    copies of the graph in as few library calls as memory allows."""
    cache_path = Path(cache_path)
    if cache_path.exists() and not force_recompute:
        return normalize_diffstg_pred_future_list(_load_tensor_list(cache_path))
    windows = list(timeseries_datas[:max_windows] if max_windows is not None else timeseries_datas)
    with torch.no_grad():
        preds, mpv, pm = _sample_stacked(model, windows, pred_len, scale=_uses_scaler(model), graph=graph_data)
    out = _DeviceCurve(preds)
    out.ews, out.pred_mean = mpv, pm
    _save_on_primary(list(out), cache_path)
    return out


def run_spatial_evaluation_cache(model, timeseries_datas, pred_len, graph_data, cache_path, device, force_recompute=False, max_windows=None):
    """NsDiff_spatial (extension: the reference's harness has no branch for it -- it would hand `evaluation_step` a tensor where
    NsDiff_model.py:695-699 reads a graph batch): windows -> list of CPU tensors [V, pred_len, F, n_z_samples], cached like run_evaluation_cache;
    every window is one copy of `graph_data`, f(x) = Model_spatial, g(x) and the sampler run for all windows in as few library calls as memory allows."""
    cache_path = Path(cache_path)
    if cache_path.exists() and not force_recompute:
        return _load_tensor_list(cache_path)
    windows = list(timeseries_datas[:max_windows] if max_windows is not None else timeseries_datas)
    with torch.no_grad():
        preds, mpv, pm = _sample_stacked(model, windows, pred_len, scale=_uses_scaler(model), graph=graph_data)
    out = _DeviceCurve(preds)
    out.ews, out.pred_mean = mpv, pm
    _save_on_primary(list(out), cache_path)
    return out


def run_nsdiff_g_cache(model, timeseries_datas, cache_path, device, pred_dim=0, force_recompute=False, max_windows=None):
    """sigma head only, one [V, pred_len, F] tensor per window (reference :400-422)."""
    cache_path = Path(cache_path)
    if cache_path.exists() and not force_recompute:
        return _load_tensor_list(cache_path)
    if getattr(model, "cond_pred_model_g", None) is None:
        return None
    windows = list(timeseries_datas[:max_windows] if max_windows is not None else timeseries_datas)
    g_list: List[torch.Tensor] = []
    if windows:
        V = int(windows[0].shape[0])
        ctx = model.native_context()
        scale = _uses_scaler(model)
        with torch.no_grad():
            rows = _stack_windows(windows).to(ctx.device)
            gx = ctx.sigma_head(rows, 1, 0, 1, model.scaler_mean if scale else None, model.scaler_std if scale else None)
        gx = gx.reshape(len(windows), V, model.pred_len, model.dataset_nf).cpu()
        if pred_dim >= gx.shape[-1]:
            raise IndexError("pred_dim {} out of bounds for F={}.".format(pred_dim, gx.shape[-1]))
        g_list = list(gx.unbind(0))
    _save_on_primary(g_list, cache_path)
    return g_list


# ----------------------------------------------------------------------------------------
# summaries
# ----------------------------------------------------------------------------------------
def _feature_inverse_transform(pred_future, model=None):
    if model is None or getattr(model, "scaler", None) is None:
        return pred_future
    if not hasattr(model, "scaler_mean") or not hasattr(model, "scaler_std"):
        return model.scaler_inverse_transform(pred_future) if hasattr(model, "scaler_inverse_transform") else pred_future
    mean = model.scaler_mean.detach().to(pred_future.device, pred_future.dtype)
    std = model.scaler_std.detach().to(pred_future.device, pred_future.dtype)
    if pred_future.ndim >= 3 and pred_future.shape[-2] == mean.numel():
        shape = [1] * pred_future.ndim
        shape[-2] = mean.numel()
        return pred_future * std.view(*shape) + mean.view(*shape)
    if pred_future.shape[-1] == mean.numel():
        return pred_future * std + mean
    return pred_future


def _reduce_on_device(tensors: Sequence[torch.Tensor], model, per_feature=False, inverse=True):
    """Stack same-shaped [V, O, F, S] windows and run the HIP variance reduction."""
    from .. import _native
    dev = None
    if model is not None and hasattr(model, "native_context"):
        ctx = model.native_context()
    else:
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        ctx = _native.Context(dev)
    x = torch.stack([torch.as_tensor(t).float() for t in tensors]).to(ctx.device).contiguous()
    std = mean = None
    if inverse and model is not None and getattr(model, "scaler", None) is not None and hasattr(model, "scaler_mean") \
            and x.shape[-2] == model.scaler_mean.numel():
        std = model.scaler_std.detach().to(ctx.device).float().contiguous()
        mean = model.scaler_mean.detach().to(ctx.device).float().contiguous()
    mpv, pm, feat = ctx.reduce(x, std, mean, per_feature=per_feature)
    return mpv.cpu().numpy(), pm.cpu().numpy(), None if feat is None else feat.cpu().numpy()


def summarize_pred_future_list(pred_future_list, model=None):
    """-> (pred_mean list, MPV list): inverse-scale, population variance over samples, mean over
    [Node, pred_len, F] (reference :286-303)."""
    if isinstance(pred_future_list, _DeviceCurve) and pred_future_list.ews is not None:
        return [np.asarray(v) for v in pred_future_list.pred_mean], [np.asarray(v) for v in pred_future_list.ews]
    items = []
    for pred_future in pred_future_list:
        pred_future = torch.as_tensor(pred_future).detach()
        if pred_future.ndim == 3:
            pred_future = pred_future.unsqueeze(0)
        if pred_future.ndim != 4:
            raise ValueError("pred_future must have shape [Node, pred_len, F, n_z_samples], got {}".format(
                tuple(pred_future.shape)))
        items.append(pred_future)
    if not items:
        return [], []
    mpv, pm, _ = _reduce_on_device(items, model)
    return [np.asarray(v) for v in pm], [np.asarray(v) for v in mpv]


def summarize_nsdiff_g_list(g_list, pred_dim=0):
    """gx-only signal: per window mean over nodes of the horizon-mean of one feature (reference :306-320)."""
    ews_list, pred_mean_list = [], []
    for gx in g_list:
        gx = torch.as_tensor(gx).detach().cpu()
        if gx.ndim == 2:
            gx = gx.unsqueeze(0)
        if gx.ndim != 3:
            raise ValueError("NsDiff-g cache elements must have shape [Node, pred_len, F].")
        if pred_dim >= gx.shape[-1]:
            raise IndexError("pred_dim {} out of bounds for F={}.".format(pred_dim, gx.shape[-1]))
        ews_list.append(gx.mean(dim=1)[:, pred_dim].mean().numpy())
        pred_mean_list.append(gx.mean().numpy())
    return pred_mean_list, ews_list


def summarize_slbp_sensitivity(pred_future_list, pred_datas, model=None, device=None, pred_dim=0):
    """SLBP MPV = mean over the horizon of the sample variance of ONE feature, no inverse scaling;
    prediction error = |sample mean - scaled truth| (reference :529-550)."""
    items = []
    for pred_future in pred_future_list:
        pred_future = torch.as_tensor(pred_future).detach()
        if pred_future.ndim != 3:
            raise ValueError("SLBP sensitivity cache elements must have shape [pred_len, F, n_z_samples].")
        if pred_dim >= pred_future.shape[1]:
            raise IndexError("pred_dim {} out of bounds for F={}.".format(pred_dim, pred_future.shape[1]))
        items.append(pred_future.unsqueeze(0))
    mpv_list = []
    if items:
        _, _, feat = _reduce_on_device(items, model, per_feature=True, inverse=False)
        mpv_list = [np.asarray(v) for v in feat[:, pred_dim]]
    err_list = []
    for pred_future, truth in zip(pred_future_list, pred_datas):
        pred_future = torch.as_tensor(pred_future).detach().cpu()
        truth = torch.as_tensor(truth).detach().cpu()
        if getattr(model, "scaler", None) is not None and hasattr(model, "scaler_transform"):
            truth = (truth - model.scaler_mean.cpu()) / model.scaler_std.cpu()
        err = torch.abs(pred_future.mean(dim=-1) - truth).mean(dim=0)
        err_list.append(err[pred_dim].numpy())
    return mpv_list, err_list


def _slbp_intrinsic_dimension(trajectories):
    """Number of principal components holding 80% of the sample-trajectory variance (reference :686-698)."""
    traj = torch.as_tensor(trajectories, dtype=torch.float32)
    if traj.ndim != 2 or traj.shape[0] < 2:
        return np.nan
    centered = traj - traj.mean(dim=0, keepdim=True)
    cov = centered.T.mm(centered) / max(traj.shape[0] - 1, 1)
    eig = torch.sort(torch.linalg.eigvalsh(cov), descending=True).values.clamp_min(0)
    total = eig.sum()
    if float(total) <= 0:
        return np.nan
    return int(torch.where(torch.cumsum(eig / total, dim=0) >= 0.8)[0][0].item() + 1)


def summarize_slbp_sampling_for_fig6(pred_future_list, pred_dim=0):
    mpv_list, _ = summarize_slbp_sensitivity(pred_future_list, (), pred_dim=pred_dim)
    dims = []
    for pred_future in pred_future_list:
        pred_future = torch.as_tensor(pred_future).detach().cpu()
        dims.append(_slbp_intrinsic_dimension(pred_future.permute(2, 0, 1).reshape(pred_future.shape[-1], -1)))
    return [float(v) for v in mpv_list], dims


# ----------------------------------------------------------------------------------------
# public entry points
# ----------------------------------------------------------------------------------------
def slbp_sensitivity_ews(model_root, model_name, torch_time_series, time_data, data_trend="increase", pred_dim=0,
                         sample_window_step=10, infer_params=None, force_recompute=False, max_windows=None, device=None):
    """SLBP single-series MPV curve for one model of a sensitivity sweep (reference :553-621)."""
    model_root = _resolve_project_path(model_root)
    device = _default_device(device)
    model, method_config, net_param = load_sensitivity_model(model_root, model_name, device=device, infer_params=infer_params)
    dataset_config = method_config.get("dataset", {})
    windows, pred_len, sampling_t = dataset_config["windows"], dataset_config["pred_len"], dataset_config["sampling_t"]
    input_datas, pred_datas, time_points = build_slbp_sensitivity_windows(
        torch_time_series, time_data, windows, pred_len, sampling_t, sample_window_step)
    cache_path = slbp_sensitivity_cache_path(model_root, model_name, data_trend, sample_window_step)
    preds = run_slbp_sensitivity_cache(model, input_datas, cache_path, device, force_recompute=force_recompute,
                                       max_windows=max_windows)
    mpv_list, err_list = summarize_slbp_sensitivity(preds, pred_datas, model=model, device=device, pred_dim=pred_dim)
    return {
        "time_points": time_points, "mpv": mpv_list, "prediction_error": err_list, "pred_future_list": preds,
        "cache_path": str(cache_path), "windows": windows, "pred_len": pred_len, "sampling_t": sampling_t,
        "sample_window_step": sample_window_step, "model_root": str(model_root), "model_name": model_name,
        "loaded_net_param": net_param,
    }


# ----------------------------------------------------------------------------------------
# SLBP figure 5 / figure 6 analyses: which cache file holds what, and how it is summarised
# (reference :624-1250).  All sampling goes through run_slbp_sensitivity_cache, all gx through the
# native sigma head; the rest is cache bookkeeping that must stay file-compatible with the reference.
# ----------------------------------------------------------------------------------------
def _as_path(path):
    return path if path is None or isinstance(path, Path) else Path(path)


def _fig6_cache_dir(model_root, cache_subdir):
    d = _resolve_project_path(model_root) / "datas"
    return d / cache_subdir if cache_subdir else d


def slbp_fig6_cache_path(model_root, model_name, data_trend, sample_window_step=10, cache_subdir=None, kind="pred_future"):
    return _fig6_cache_dir(model_root, cache_subdir) / "{}_{}_{}_{}.pt".format(model_name, kind, data_trend, sample_window_step)


def slbp_fig6_pred_future_gx_cache_path(model_root, model_name, data_trend, sample_window_step=10, cache_subdir=None):
    return _fig6_cache_dir(model_root, cache_subdir) / "{}_pred_future_{}_{}_gx.pt".format(model_name, data_trend, sample_window_step)


def _legacy_single_underscore_model_name(model_name):
    return str(model_name).replace("dataset__", "dataset_", 1)


def _read_slbp_fig6_model_config(model_root, model_name):
    config_path = _resolve_project_path(model_root) / "models" / "{}.yaml".format(model_name)
    if not config_path.exists():
        return None
    with open(config_path, "r", encoding="utf-8") as f:
        return yaml.safe_load(f)


def _slbp_fig6_dataset_config(model_root, model_name, windows=None, pred_len=None, sampling_t=None):
    method_config = _read_slbp_fig6_model_config(model_root, model_name)
    ds = method_config.get("dataset", {}) if method_config else {}
    pick = lambda given, key, default: given if given is not None else ds.get(key, default)
    return {"windows": pick(windows, "windows", 200), "pred_len": pick(pred_len, "pred_len", 200),
            "sampling_t": pick(sampling_t, "sampling_t", 100), "method_config": method_config}


def _slbp_cache_elements_have_ndim(data_list, ndim):
    return bool(data_list) and all(torch.as_tensor(item).ndim == ndim for item in data_list)


def _slbp_cache_elements_are_gx(data_list):
    """[pred_len, F] (or [1, pred_len, F]) elements = a sigma-head cache; [pred_len, F, S] = a sampling cache."""
    if _slbp_cache_elements_have_ndim(data_list, 2):
        return True
    return bool(data_list) and all(torch.as_tensor(i).ndim == 3 and torch.as_tensor(i).shape[0] == 1 for i in data_list)


def summarize_slbp_gx_for_fig6(gx_list, pred_dim=0):
    out = []
    for gx in gx_list:
        gx = torch.as_tensor(gx).detach().cpu()
        if gx.ndim == 3 and gx.shape[0] == 1:
            gx = gx.squeeze(0)
        if gx.ndim != 2:
            raise ValueError("SLBP gx cache elements must have shape [pred_len, F] or [1, pred_len, F].")
        if pred_dim >= gx.shape[-1]:
            raise IndexError("pred_dim {} out of bounds for F={}.".format(pred_dim, gx.shape[-1]))
        out.append(float(gx[:, pred_dim].mean().numpy()))
    return out


def run_slbp_gx_cache_for_fig6(model, input_datas, cache_path, device, pred_dim=0, force_recompute=False, max_windows=None):
    """sigma head over single-series windows -> list of [pred_len, F] (reference :731-765), one launch."""
    cache_path = Path(cache_path)
    if cache_path.exists() and not force_recompute:
        cached = _load_tensor_list(cache_path)
        if _slbp_cache_elements_are_gx(cached):
            return cached
    if getattr(model, "cond_pred_model_g", None) is None:
        raise ValueError("model does not provide cond_pred_model_g for gx generation.")
    windows = [torch.as_tensor(w).float().unsqueeze(0) for w in
               (input_datas[:max_windows] if max_windows is not None else input_datas)]
    gx_list = []
    if windows:
        ctx = model.native_context()
        scale = _uses_scaler(model)
        with torch.no_grad():
            gx = ctx.sigma_head(_stack_windows(windows).to(ctx.device), 1, 0, 1, model.scaler_mean if scale else None,
                                model.scaler_std if scale else None)
        gx = gx.reshape(len(windows), model.pred_len, model.dataset_nf).cpu()
        if pred_dim >= gx.shape[-1]:
            raise IndexError("pred_dim {} out of bounds for F={}.".format(pred_dim, gx.shape[-1]))
        gx_list = list(gx.unbind(0))
    _save_on_primary(gx_list, cache_path)
    return gx_list


def _fig6_windows(cfg, torch_time_series, time_data, sample_window_step):
    return build_slbp_sensitivity_windows(torch_time_series, time_data, cfg["windows"], cfg["pred_len"], cfg["sampling_t"],
                                          sample_window_step)


def _cfg_fields(cfg, sample_window_step):
    return {"windows": cfg["windows"], "pred_len": cfg["pred_len"], "sampling_t": cfg["sampling_t"],
            "sample_window_step": sample_window_step}


def slbp_sampling_analysis(model_root, model_name, torch_time_series, time_data, data_trend="increase", pred_dim=0,
                           sample_window_step=10, cache_subdir=None, windows=None, pred_len=None, sampling_t=None,
                           infer_params=None, force_recompute=False, max_windows=None, device=None, allow_unavailable=True):
    """Figure-6 sampling curve: MPV and PCA dimension per window (reference :768-885).  The shared
    `pred_future` cache name may hold a gx cache from older runs; sampling then lives in `sampling_pred_future`."""
    cfg = _slbp_fig6_dataset_config(model_root, model_name, windows, pred_len, sampling_t)
    input_datas, _targets, time_points = _fig6_windows(cfg, torch_time_series, time_data, sample_window_step)
    shared = slbp_fig6_cache_path(model_root, model_name, data_trend, sample_window_step, cache_subdir, "pred_future")
    own = slbp_fig6_cache_path(model_root, model_name, data_trend, sample_window_step, cache_subdir, "sampling_pred_future")
    try:
        active, preds = shared, None
        if not force_recompute:
            if shared.exists():
                cand = _load_tensor_list(shared)
                if _slbp_cache_elements_have_ndim(cand, 3):
                    preds = cand
                else:
                    active = own
            if preds is None and own.exists():
                cand = _load_tensor_list(own)
                if not _slbp_cache_elements_have_ndim(cand, 3):
                    raise ValueError("sampling cache exists but is not [pred_len, F, n_z_samples]: {}".format(own))
                preds, active = cand, own
        if preds is None:
            if shared.exists() and active == shared:
                active = own
            device = _default_device(device)
            model, _cfg, _np = load_sensitivity_model(model_root, model_name, device=device, infer_params=infer_params)
            preds = run_slbp_sensitivity_cache(model, input_datas, active, device, force_recompute=force_recompute,
                                               max_windows=max_windows)
            if not _slbp_cache_elements_have_ndim(preds, 3):
                raise ValueError("generated sampling cache is not [pred_len, F, n_z_samples]: {}".format(active))
        mpv, dims = summarize_slbp_sampling_for_fig6(preds, pred_dim=pred_dim)
        return dict(available=True, time_points=time_points[: len(mpv)], mpv=mpv, intrinsic_dimension=dims,
                    pred_future_list=preds, cache_path=str(active), reason="", **_cfg_fields(cfg, sample_window_step))
    except Exception as exc:
        if not allow_unavailable:
            raise
        return dict(available=False, time_points=time_points, mpv=[], intrinsic_dimension=[], pred_future_list=None,
                    cache_path=str(own if shared.exists() else shared), reason=str(exc),
                    **_cfg_fields(cfg, sample_window_step))


def slbp_gx_analysis(model_root, model_name, torch_time_series, time_data, data_trend="increase", pred_dim=0,
                     sample_window_step=10, cache_subdir=None, windows=None, pred_len=None, sampling_t=None,
                     infer_params=None, force_recompute=False, max_windows=None, device=None):
    """Figure-6 sigma-head curve (reference :888-1023).  Cache lookup order: `<name>_pred_future_*_gx.pt`, the
    old `gx` kind, then the shared `pred_future` name (current and legacy single-underscore model name) if
    it happens to hold a gx cache; otherwise compute into the first."""
    cfg = _slbp_fig6_dataset_config(model_root, model_name, windows, pred_len, sampling_t)
    input_datas, _targets, time_points = _fig6_windows(cfg, torch_time_series, time_data, sample_window_step)
    target = slbp_fig6_pred_future_gx_cache_path(model_root, model_name, data_trend, sample_window_step, cache_subdir)
    old_kind = slbp_fig6_cache_path(model_root, model_name, data_trend, sample_window_step, cache_subdir, "gx")
    shared = [slbp_fig6_cache_path(model_root, n, data_trend, sample_window_step, cache_subdir, "pred_future")
              for n in (model_name, _legacy_single_underscore_model_name(model_name))]

    def compute():
        dev = _default_device(device)
        model, _cfg, _np = load_sensitivity_model(model_root, model_name, device=dev, infer_params=infer_params)
        return run_slbp_gx_cache_for_fig6(model, input_datas, target, dev, pred_dim=pred_dim, force_recompute=force_recompute,
                                          max_windows=max_windows)

    used = target
    if force_recompute:
        gx_list = compute()
    elif target.exists():
        gx_list = _load_tensor_list(target)
    elif old_kind.exists():
        gx_list = _load_tensor_list(old_kind)
        if not _slbp_cache_elements_are_gx(gx_list):
            raise ValueError("gx cache exists but is not a gx cache: {}".format(old_kind))
        used = old_kind
    else:
        first = next((c for c in shared if c.exists()), None)    # only the first existing candidate is examined
        cand = _load_tensor_list(first) if first is not None else None
        if cand is not None and _slbp_cache_elements_are_gx(cand):
            gx_list, used = cand, first
        else:
            gx_list = compute()
    return dict(time_points=time_points[: len(gx_list)], gx_mpv=summarize_slbp_gx_for_fig6(gx_list, pred_dim=pred_dim),
                gx_list=gx_list, cache_path=str(used), **_cfg_fields(cfg, sample_window_step))


def summarize_slbp_mpv_cache_for_fig5(data_list, pred_dim=0):
    """-> (curve, "sampling" | "gx") according to what the cache elements look like (reference :1026-1036)."""
    if _slbp_cache_elements_have_ndim(data_list, 3):
        return summarize_slbp_sampling_for_fig6(data_list, pred_dim=pred_dim)[0], "sampling"
    if _slbp_cache_elements_are_gx(data_list):
        return summarize_slbp_gx_for_fig6(data_list, pred_dim=pred_dim), "gx"
    shape = tuple(torch.as_tensor(data_list[0]).shape) if data_list else None
    raise ValueError("Unsupported SLBP MPV cache element shape: {}".format(shape))


def _cache_time_points(time_data, sampling_t, windows, cache_len, fallback_step, n):
    sampled_time = torch_data_preprocessing_like_slbp(time_data, sampling_t=sampling_t, return_numpy=True)
    step = infer_sample_window_step_from_cache(sampled_length=len(sampled_time), windows=windows, cache_len=cache_len,
                                               fallback_step=fallback_step)
    return sampled_time[windows - 1:: step][:n], step


def slbp_mpv_analysis(model_root, model_name, torch_time_series, time_data, cache_path, pred_dim=0, sample_window_step=10,
                      windows=None, pred_len=None, sampling_t=None, infer_params=None, force_recompute=False,
                      max_windows=None, device=None):
    """Figure-5 curve from an explicit cache file of either kind; the window step of an existing cache is
    inferred from its length (reference :1039-1114)."""
    cfg = _slbp_fig6_dataset_config(model_root, model_name, windows, pred_len, sampling_t)
    cache_path = _resolve_project_path(cache_path)
    if cache_path.exists() and not force_recompute:
        data_list = _load_tensor_list(cache_path)
        mpv, source = summarize_slbp_mpv_cache_for_fig5(data_list, pred_dim=pred_dim)
        tp, step = _cache_time_points(time_data, cfg["sampling_t"], cfg["windows"], len(data_list), sample_window_step, len(mpv))
    else:
        dev = _default_device(device)
        model, _cfg, _np = load_sensitivity_model(model_root, model_name, device=dev, infer_params=infer_params)
        input_datas, _targets, time_points = _fig6_windows(cfg, torch_time_series, time_data, sample_window_step)
        data_list = run_slbp_sensitivity_cache(model, input_datas, cache_path, dev, force_recompute=force_recompute,
                                               max_windows=max_windows)
        mpv, source = summarize_slbp_mpv_cache_for_fig5(data_list, pred_dim=pred_dim)
        tp, step = time_points[: len(mpv)], sample_window_step
    return dict(time_points=tp, mpv=mpv, pred_future_list=data_list, cache_path=str(cache_path), uncertainty_source=source,
                **_cfg_fields(cfg, step))


def slbp_direct_model_cache_analysis(model_save_file, torch_time_series, time_data, cache_path, pred_dim=0,
                                     sample_window_step=10, cache_kind="auto", infer_params=None, force_recompute=False,
                                     max_windows=None, device=None, compute_prediction_error=False):
    """Same as slbp_mpv_analysis for a model directory (model_trained + .yaml) instead of a sweep root
    (reference :1117-1230); optionally adds the prediction error of a sampling cache."""
    method_config = read_model_config(model_save_file)
    ds, net = method_config.get("dataset", {}), method_config.get("net", {})
    windows = int(ds.get("windows", net.get("windows", 200)))
    pred_len = int(ds.get("pred_len", net.get("pred_len", 200)))
    sampling_t = int(ds.get("sampling_t", 100))
    cache_path = _resolve_project_path(cache_path)
    model = None

    def need_model():
        nonlocal model, device
        if model is None:
            device = _default_device(device)
            model, _np = load_model_from_dir(model_save_file, device=device, infer_params=infer_params, method_config=method_config)
        return model

    if cache_path.exists() and not force_recompute:
        data_list = _load_tensor_list(cache_path)
    else:
        if cache_kind not in {"gx", "sampling"}:
            raise ValueError("cache_kind must be 'gx' or 'sampling' when cache is missing.")
        input_datas, _targets, _tp = build_slbp_sensitivity_windows(torch_time_series, time_data, windows, pred_len, sampling_t,
                                                                   sample_window_step)
        runner = run_slbp_gx_cache_for_fig6 if cache_kind == "gx" else run_slbp_sensitivity_cache
        extra = {"pred_dim": pred_dim} if cache_kind == "gx" else {}
        data_list = runner(need_model(), input_datas, cache_path, device, force_recompute=force_recompute,
                           max_windows=max_windows, **extra)
    mpv, source = summarize_slbp_mpv_cache_for_fig5(data_list, pred_dim=pred_dim)
    tp, step = _cache_time_points(time_data, sampling_t, windows, len(data_list), sample_window_step, len(mpv))
    result = dict(time_points=tp, mpv=mpv, pred_future_list=data_list, cache_path=str(cache_path), windows=windows,
                  pred_len=pred_len, sampling_t=sampling_t, sample_window_step=step, uncertainty_source=source)
    if compute_prediction_error:
        if source != "sampling":
            raise ValueError("prediction_error requires a sampling cache, got '{}'.".format(source))
        _inp, targets, _tp = build_slbp_sensitivity_windows(torch_time_series, time_data, windows, pred_len, sampling_t, step)
        _mpv, result["prediction_error"] = summarize_slbp_sensitivity(data_list, targets[: len(data_list)], model=need_model(),
                                                                      device=device, pred_dim=pred_dim)
    return result


def slbp_raw_window_variance(torch_time_series, time_data, windows=200, sampling_t=100, sample_window_step=10, pred_dim=0):
    """The classical indicator beside the MPV curve: variance of each raw sliding window (reference :1233-1250)."""
    series = torch_data_preprocessing_like_slbp(torch_time_series, sampling_t=sampling_t)
    sampled_time = torch_data_preprocessing_like_slbp(time_data, sampling_t=sampling_t, return_numpy=True)
    if series.ndim != 2:
        raise ValueError("SLBP raw series must have shape [T, F].")
    if pred_dim >= series.shape[1]:
        raise IndexError("pred_dim {} out of bounds for F={}.".format(pred_dim, series.shape[1]))
    variances = series[:, pred_dim].unfold(0, windows, sample_window_step).var(dim=1, unbiased=False).cpu().numpy()
    return {"time_points": sampled_time[windows - 1:: sample_window_step][: len(variances)], "variance": variances,
            "windows": windows, "sampling_t": sampling_t, "sample_window_step": sample_window_step}


_METHOD_ALIASES = {"variance": "sampling", "sampling_variance": "sampling", "pred_future": "sampling", "pred": "sampling",
                   "g": "gx", "preg": "gx", "nsdiff_g": "gx"}


def uncertainty_ews(model_save_file=None, data_file=None, torch_time_series=None, time_data=None, dynamic_type=None,
                    task_model=None, graph_file=None, cache_path=None, sample_window_step=None, sampling_t=None,
                    infer_params=None, pred_dim=0, force_recompute=False, save_nsdiff_g=True, nsdiff_g_path=None,
                    uncertainty_method="sampling", max_windows=None, device=None, load_model_when_cached=False,
                    store_samples=True):
    """Trajectory + trained model -> MPV early-warning curve (reference :1253-1541).

    Same arguments and result keys as the reference.  `uncertainty_method`: "sampling" (variance
    of N sampled futures), "gx" (sigma head only) or "both".  Extension: store_samples=False computes the curve only (the
    MPV reduction runs in the sampler's epilogue, no sample tensor, no cache file, `pred_future_list` is None).
    """
    dynamic_type = _dynamic_name(dynamic_type)
    method = _METHOD_ALIASES.get(str(uncertainty_method).lower(), str(uncertainty_method).lower())
    if method not in {"sampling", "gx", "both"}:
        raise ValueError("uncertainty_method must be one of: sampling, gx, both.")

    if data_file is not None:
        data_file = _resolve_project_path(data_file)
        loaded = load_dynamic_data(data_file, dynamic_type=dynamic_type)
        torch_time_series, time_data = loaded["torch_time_series"], loaded["time_data"]
        dynamic_type = dynamic_type or loaded["dynamic_type"]
    elif torch_time_series is not None:
        torch_time_series = normalize_time_series(torch_time_series, dynamic_type=dynamic_type)
    else:
        raise ValueError("Provide data_file or torch_time_series.")
    if time_data is None:
        raise ValueError("time_data is required when data_file is not provided.")

    method_config = None
    if model_save_file is not None:
        model_save_file = _resolve_project_path(model_save_file)
        method_config = read_model_config(model_save_file)
    if task_model is None and method_config is not None:
        task_model = method_config.get("net", {}).get("task_model")
    dataset_config = method_config.get("dataset", {}) if method_config else {}
    windows, pred_len = dataset_config.get("windows"), dataset_config.get("pred_len")
    if windows is None or pred_len is None:
        raise ValueError("model_trained.yaml must provide dataset.windows and dataset.pred_len.")

    cache_path = resolve_cache_path(cache_path, model_save_file, data_file, dynamic_type)
    want_sampling = method in {"sampling", "both"}
    want_gx = method in {"gx", "both"} or (save_nsdiff_g and method == "sampling")
    gx_path = None
    if want_gx:
        gx_path = resolve_cache_path(nsdiff_g_path if nsdiff_g_path is not None else cache_path.parent, model_save_file,
                                     data_file, dynamic_type, suffix="_gx")
    cached_preds = _load_tensor_list(cache_path) if want_sampling and cache_path.exists() and not force_recompute else None
    cached_gx = _load_tensor_list(gx_path) if want_gx and gx_path is not None and gx_path.exists() and not force_recompute else None

    if sampling_t is None:
        sampling_t = dataset_config.get("sampling_t", DEFAULT_SAMPLING_T.get(dynamic_type, 0.1))
    sampled_series, sampled_time = sample_time_series(torch_time_series, time_data, sampling_t=sampling_t)
    if sample_window_step is None:
        fallback = default_sample_window_step(dynamic_type, task_model=task_model, dataset_config=dataset_config)
        known = len(cached_preds) if cached_preds is not None else (len(cached_gx) if cached_gx is not None else None)
        sample_window_step = fallback if known is None else infer_sample_window_step_from_cache(
            sampled_series.shape[1], windows, known, fallback)
    window_views, time_points = build_sliding_windows(sampled_series, sampled_time, windows, sample_window_step)

    model, net_param = None, None

    def need_model():
        nonlocal model, net_param, device
        if model is None:
            if model_save_file is None:
                raise ValueError("model_save_file is required when cache_path does not exist or force_recompute=True.")
            device = _default_device(device)
            model, net_param = load_model_from_dir(model_save_file, device=device, infer_params=infer_params,
                                                   method_config=method_config)
        return model

    preds, pred_mean_list, ews_list = None, [], []
    if want_sampling and task_model == "DiffSTG":                       # reference :1393-1430
        if dynamic_type not in NETWORK_DYNAMICS:
            raise ValueError("DiffSTG only supports network dynamics: SIS, neuronal, biomass.")
        if graph_file is None:
            raise ValueError("graph_file is required for DiffSTG.")
        if cached_preds is not None:
            preds = normalize_diffstg_pred_future_list(cached_preds)
            if model_save_file is not None and load_model_when_cached:
                need_model()
        else:
            if infer_params is None:
                infer_params = {"parallel_sampling": 10, "sequential_sampling": 1, "n_z_samples": 10, "diffusion_steps": 20}
            preds = run_diffstg_evaluation_cache(need_model(), window_views, pred_len, load_diffstg_graph(graph_file), cache_path, device,
                                                 force_recompute=force_recompute, max_windows=max_windows)
        pred_mean_list, ews_list = summarize_pred_future_list(preds, model=model)
    elif want_sampling and task_model == "NsDiff_spatial":             # extension (run_spatial_evaluation_cache)
        if dynamic_type not in NETWORK_DYNAMICS:
            raise ValueError("NsDiff_spatial only supports network dynamics: SIS, neuronal, biomass.")
        if graph_file is None:
            raise ValueError("graph_file is required for NsDiff_spatial.")
        if cached_preds is not None:
            preds = cached_preds
            if model_save_file is not None and load_model_when_cached:
                need_model()
        else:
            preds = run_spatial_evaluation_cache(need_model(), window_views, pred_len, load_diffstg_graph(graph_file), cache_path, device,
                                                 force_recompute=force_recompute, max_windows=max_windows)
        pred_mean_list, ews_list = summarize_pred_future_list(preds, model=model)
    elif want_sampling:
        if cached_preds is not None:
            preds = cached_preds
            if model_save_file is not None and load_model_when_cached:
                need_model()
        else:
            preds = run_evaluation_cache(need_model(), window_views, pred_len, cache_path, device,
                                         force_recompute=force_recompute, max_windows=max_windows, store_samples=store_samples)
        pred_mean_list, ews_list = summarize_pred_future_list(preds, model=model)

    result = {
        "pred_future_list": list(preds) if (preds is not None and (store_samples or len(preds))) else None,
        "pred_mean": pred_mean_list, "ews": ews_list, "time_points": time_points[: len(ews_list)],
        "cache_path": str(cache_path), "figure_path": str(resolve_figure_path(cache_path)),
        "torch_time_series": torch_time_series, "time_data": torch.as_tensor(time_data).cpu().detach().numpy(),
        "dynamic_type": dynamic_type, "sampling_t": sampling_t, "sample_window_step": sample_window_step,
        "windows": windows, "pred_len": pred_len, "task_model": task_model, "uncertainty_method": method,
        "uncertainty_source": "sampling" if want_sampling else None,
        "graph_file": str(_resolve_project_path(graph_file)) if graph_file is not None else None,
        "model_save_file": str(model_save_file) if model_save_file is not None else None,
        "loaded_net_param": net_param,
    }

    has_g_model = model is not None and getattr(model, "cond_pred_model_g", None) is not None
    g_list = None
    if want_gx and ("NsDiff" in str(task_model) or has_g_model):
        if cached_gx is not None:
            g_list = cached_gx
        else:
            if model is None and model_save_file is not None:
                need_model()
                result["loaded_net_param"] = net_param
            if model is not None and getattr(model, "cond_pred_model_g", None) is not None:
                g_list = run_nsdiff_g_cache(model, window_views, gx_path, device, pred_dim=pred_dim,
                                            force_recompute=force_recompute, max_windows=max_windows)
        if g_list is not None:
            g_mean, g_ews = summarize_nsdiff_g_list(g_list, pred_dim=pred_dim)
            result["nsdiff_g"] = {"pred_future_list": g_list, "pred_mean": g_mean, "ews": g_ews,
                                  "time_points": time_points[: len(g_ews)], "cache_path": str(gx_path)}
            if method == "gx":
                result.update({"pred_future_list": None, "pred_mean": g_mean, "ews": g_ews,
                               "time_points": time_points[: len(g_ews)], "cache_path": str(gx_path),
                               "figure_path": str(resolve_figure_path(gx_path)), "uncertainty_source": "gx"})
    if method == "gx" and g_list is None:
        raise ValueError("uncertainty_method='gx' requires a task_model containing 'NsDiff' "
                         "and a loaded model with cond_pred_model_g, or an existing _gx cache.")
    return result
