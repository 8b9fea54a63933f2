"""NsDiff model objects -- the reference's duck-typed "operator API" for the MPV path.

Mirrors models/Diffusion_model/NsDiff/NsDiff_model.py of the reference:
`NsDiff_model_variants` (:272-495) and the g-only / full `NsDiff_model` (:17-268).  Same
constructor arguments, attribute names (`scaler`, `scaler_mean`, `scaler_std`, `model`,
`cond_pred_model`, `cond_pred_model_g`, `configs`), parameter names and
`evaluation_step(batch) -> (outs [B, pred_len, F, n_z_samples] on CPU, batch_y | None)`.

What differs is where the arithmetic runs: `evaluation_step` hands the window batch to the
HIP library (sigma head -> fused T-step chain sampler), and `sample_windows` does the same
for many sliding windows of a resident series in ONE launch, which is what the harness
uses.  Nothing here computes the path with torch ops.

Noise: `noise_source = "philox"` (default) draws on device (Philox4x32-10, seed taken from
torch's CPU generator so `torch.manual_seed` makes runs repeatable); `"torch_cpu"` draws
every normal with `torch.randn` on the host in the reference's exact order
 and uploads it, which reproduces a reference CPU run under the same
`torch.manual_seed` to fp32 round-off.  Select with net_param / infer_params key
`noise_source`, or pass `noise=` explicitly.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from . import g_backbone as G
from . import mu_backbone as ns_Transformer
from .NsDiff_net import NsDiff_net, NsDiff_net_spatial
from .native_handle import NativeHandle

_F_BACKBONE_MSG = (
    "train_model_select={!r} needs the non-stationary-transformer mean backbone f(x) (reference NsDiff/mu_backbone.py), and "
    "net_param lacks its hyper-parameters (" + ", ".join(ns_Transformer.CONFIG_KEYS) + "), so none was built.  Add them, or "
    "pass a precomputed y_0_hat (`y0hat=`), or attach your own module with the reference's call signature "
    "`(batch_x, dec_inp) -> (y_0_hat, _)` via `attach_backbone(module)`.")


def _draw_seed() -> int:
    """One 62-bit Philox key from torch's default CPU generator (repeatable under manual_seed)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class _NsDiffBase(nn.Module):
    EPS = 10e-8   # (sic) reference NsDiff_model.py:39

    def _init_common(self, net_param):
        self.device = net_param["device"]
        self.dataset_nf = net_param["dataset_nf"]
        self.windows = net_param["windows"]
        self.pred_len = net_param["pred_len"]
        self.rolling_length = net_param["rolling_length"]
        self.seq_len = net_param["seq_len"] = self.windows
        self.label_len = net_param["label_len"] = self.windows // 2
        self.diffusion_steps = net_param["diffusion_steps"]
        self.configs = SimpleNamespace(**net_param)
        self.scaler = net_param["scaler_type"]
        self.noise_source = net_param.get("noise_source", "philox")
        # "f16x3" (split-f16 matrix cores, fp32-equivalent) | "f32" (exact) | "f16x2" / "f16" (reduced-precision operands)
        self.gemm_mode = net_param.get("gemm_mode", "f16x3")
        if self.noise_source not in ("philox", "torch_cpu"):
            raise ValueError("noise_source must be 'philox' or 'torch_cpu'")
        self.register_buffer("scaler_mean", torch.zeros(self.dataset_nf))
        self.register_buffer("scaler_std", torch.zeros(self.dataset_nf))
        object.__setattr__(self, "_native", NativeHandle())
        self._native.bind(self)

    # ---- f(x): the native ns-Transformer (mu_backbone.Model -> csrc/nst.hip) when net_param carries its hyper-parameters,
    #      otherwise a user-supplied torch module ------------------------------------------------------------------
    needs_backbone = False

    def _build_backbone(self, net_param):
        if all(k in net_param for k in ns_Transformer.CONFIG_KEYS):
            return ns_Transformer.Model(self.configs, native=self._native)
        return None

    def _native_backbone(self) -> bool:
        return isinstance(getattr(self, "cond_pred_model", None), ns_Transformer.Model)

    def attach_backbone(self, module: nn.Module):
        """Install f(x) as `cond_pred_model` (same attribute name as the reference, so its checkpoint entries
        `cond_pred_model.*` load).  Call signature: module(batch_x [B, windows, F], dec_inp [B, label_len+pred_len, F])
        -> (y_0_hat [B, pred_len, F], anything)  (reference NsDiff_model.py:222, mu_backbone.py:150-183)."""
        self.cond_pred_model = module
        return self

    def _decoder_input(self, batch_x):
        zeros = torch.zeros(batch_x.size(0), self.pred_len, self.dataset_nf, device=batch_x.device, dtype=batch_x.dtype)
        return torch.cat([batch_x[:, -self.label_len:, :], zeros], dim=1)                  # reference :195-199

    def _backbone_mean(self, batch_x):
        """y_0_hat of already scaled windows [B, windows, F] through the backbone (native, or the caller's attached module)."""
        if getattr(self, "cond_pred_model", None) is None:
            raise NotImplementedError(_F_BACKBONE_MSG.format(getattr(self, "train_model_select", None)))
        if self._native_backbone():
            return self.cond_pred_model(batch_x)[0]
        dev = next(self.cond_pred_model.parameters(), torch.zeros(0, device=self.scaler_mean.device)).device
        with torch.no_grad():
            y0hat, _ = self.cond_pred_model(batch_x.to(dev), self._decoder_input(batch_x.to(dev)))
        return y0hat

    # ---- scaler (reference :100-110) ------------------------------------------------
    def scaler_fit(self, data):
        std = data.std(axis=0)
        std[std == 0] = 1
        self.scaler_mean = data.mean(axis=0)
        self.scaler_std = std

    def scaler_transform(self, data):
        return (data - self.scaler_mean) / self.scaler_std

    def scaler_inverse_transform(self, data):
        return (data * self.scaler_std) + self.scaler_mean

    # ---- keep the device copy of the weights coherent ------------------------------------
    def load_state_dict(self, state_dict, *args, **kwargs):
        """Strict by default, like the reference loader.  Entries of an f(x) backbone (`cond_pred_model.*`) are
        dropped while no backbone is attached: the sampler and sigma head load, f(x) stays the caller's business."""
        if getattr(self, "cond_pred_model", None) is None:
            state_dict = {k: v for k, v in state_dict.items() if not k.startswith("cond_pred_model.")}
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self._native.mark_dirty()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._native.mark_dirty()
        return out

    def refresh_native(self):
        """Call after editing parameters in place."""
        self._native.mark_dirty()

    def _upload_native(self, ctx):
        if getattr(self, "model", None) is not None:
            w = self.model.native_weights()
            ctx.load_nsdiff(self.diffusion_steps, self.dataset_nf, w["lin_w"], w["lin_b"], w["embed"], w["eps_w"],
                            w["eps_b"], w["sig_w"], w["sig_b"], w["coef"])
        g = getattr(self, "cond_pred_model_g", None)
        if g is not None:
            ctx.load_sigma(g.seq_len, g.kernel_size, g.pred_len, g.enc_in, g.hidden_size, g.native_weights())
        if self._native_backbone():
            ctx.load_nst(self.cond_pred_model)

    def native_context(self):
        return self._native.context()

    # ---- noise in the reference's draw order ---------------------------------------------
    def host_noise(self, n_windows: int, nodes: int) -> torch.Tensor:
        """All normals of `n_windows` evaluation_step calls, drawn with torch.randn on the CPU in
        the reference's order: window, chunk, draw k=0..T-1, tensor [nodes*P, O, F]."""
        P, S = int(self.configs.parallel_sample), int(self.configs.n_z_samples)
        draw = (nodes * P, self.pred_len, self.dataset_nf)
        # one torch.randn call per reference draw: a single big call is NOT the same stream when a
        # draw's element count is not a multiple of 16 (torch's vectorised normal fill)
        out = torch.empty((n_windows, S // P, self.diffusion_steps) + draw)
        for w in range(n_windows):
            for c in range(S // P):
                for k in range(self.diffusion_steps):
                    out[w, c, k] = torch.randn(draw)
        return out

    def _prior_variance(self, ctx, series, n_windows, first, step, scaled_input: bool):
        raise NotImplementedError

    def _check_sampling_config(self):
        P, S = int(self.configs.parallel_sample), int(self.configs.n_z_samples)
        if P < 1 or S < 1 or S % P != 0:
            # the reference floors n_z_samples // parallel_sample and then trips its shape assert
            raise AssertionError("n_z_samples ({}) must be a positive multiple of parallel_sample ({})".format(S, P))
        return S, P

    def sample_windows(self, series: torch.Tensor, n_windows: int, first: int = 0, step: int = 1, *,
                       scale_input: bool = False, y0hat: Optional[torch.Tensor] = None,
                       noise: Optional[torch.Tensor] = None, seed: Optional[int] = None, offset: int = 0,
                       summary: bool = False, want_samples: bool = True, inverse_scale: bool = True):
        """Batched device path: series [V, L, F] (resident, sub-sampled) -> samples
        [n_windows, V, pred_len, F, n_z_samples] on device, plus the gx it used.

        Window w covers series[:, first + w*step : first + w*step + windows].  With
        scale_input the scaler is fused into the load (raw series in); otherwise the series
        must already be scaled.  One sigma-head launch and one chain launch in total.

        summary=True also returns the per-window MPV and prediction mean (the reduction fused into the sampler's
        epilogue; inverse-scaled with the model's scaler when `inverse_scale` and the model has one, as
        summarize_pred_future_list does): (samples | None, gx, mpv [W], pred_mean [W]).  want_samples=False skips the
        sample tensor (curve-only runs).
        """
        if getattr(self, "model", None) is None:
            raise RuntimeError("this model has no diffusion network (train_model_select={!r})".format(
                getattr(self, "train_model_select", None)))
        S, P = self._check_sampling_config()
        ctx = self._native.context()
        series = series.to(device=ctx.device, dtype=torch.float32).contiguous()
        gx = self._prior_variance(ctx, series, n_windows, first, step, scale_input)
        if y0hat is None and self.needs_backbone and self._native_backbone():        # windows gathered by the kernel, scaler fused
            y0hat = ctx.nst_forward(series, n_windows, first, step, self.scaler_mean if scale_input else None,
                                    self.scaler_std if scale_input else None, gemm=self.cond_pred_model.gemm_mode)
        elif y0hat is None and self.needs_backbone:
            idx = first + step * torch.arange(n_windows, device=series.device).view(-1, 1) + torch.arange(self.windows, device=series.device)
            wins = series[:, idx, :].permute(1, 0, 2, 3).reshape(n_windows * series.shape[0], self.windows, -1)
            if scale_input:
                wins = (wins - self.scaler_mean.to(wins.device)) / self.scaler_std.to(wins.device)
            y0hat = self._backbone_mean(wins)
        if y0hat is not None:
            y0hat = y0hat.to(device=ctx.device, dtype=torch.float32).contiguous().reshape(gx.shape)
        if noise is None and self.noise_source == "torch_cpu":
            noise = self.host_noise(n_windows, series.shape[0])
        if noise is not None:
            noise = noise.to(device=ctx.device, dtype=torch.float32).contiguous()
        elif seed is None:
            seed = _draw_seed()
        if not summary:
            samples = ctx.sample(gx, y0hat, S, P, noise=noise, seed=seed or 0, offset=offset, gemm=self.gemm_mode)
            return samples, gx
        std = mean = None
        if inverse_scale and self.scaler is not None:
            std = self.scaler_std.to(ctx.device).float().contiguous()
            mean = self.scaler_mean.to(ctx.device).float().contiguous()
        samples, mpv, pm, _ = ctx.sample(gx, y0hat, S, P, noise=noise, seed=seed or 0, offset=offset, gemm=self.gemm_mode,
                                         summary=True, want_samples=want_samples, std=std, mean=mean)
        return samples, gx, mpv, pm

    def _evaluation_step(self, batch, y0hat, noise):
        if batch.dim() != 3 or batch.shape[1] < self.windows:
            raise ValueError("batch must be [B, >= windows ({}), F], got {}".format(self.windows, tuple(batch.shape)))
        batch_y = None
        if batch.shape[1] - self.windows >= self.pred_len:
            batch_y = batch[:, self.windows:, :].to(self.device)
            assert batch_y.size(1) == self.pred_len, "pred_len is not equal to the length of the prediction"
        x = batch[:, : self.windows, :]
        if noise is not None and noise.dim() != 6:
            noise = noise.reshape(1, -1, self.diffusion_steps, x.shape[0] * int(self.configs.parallel_sample),
                                  self.pred_len, self.dataset_nf)
        samples, _ = self.sample_windows(x, 1, 0, 1, y0hat=y0hat, noise=noise)
        outs = samples[0].cpu()                                   # [B, O, F, S], the reference's `outs`
        assert tuple(outs.shape[1:]) == (self.pred_len, self.dataset_nf, int(self.configs.n_z_samples))
        return outs, batch_y


class NsDiff_model_variants(_NsDiffBase):
    """Ablation family of the reference (:272-495).  'cond_var' (sigma head, zero mean) and
    'Guassian' (sic: unit variance, zero mean) need no f(x) backbone and are fully native."""

    def __init__(self, net_param, train_model_select):
        super().__init__()
        self._init_common(net_param)
        self.train_model_select = train_model_select
        if train_model_select not in ("Guassian", "cond_mean", "cond_var", "wo_UANS"):
            raise ValueError("train_model_select should be in Guassian/cond_mean/cond_var")
        self.needs_backbone = train_model_select in ("cond_mean", "wo_UANS")      # f(x): attach_backbone() or y0hat=
        self.model = NsDiff_net(self.configs, self.device)
        self.cond_pred_model = self._build_backbone(net_param) if self.needs_backbone else None      # reference :299-311
        self.cond_pred_model_g = None
        if train_model_select in ("cond_var", "wo_UANS"):
            self.cond_pred_model_g = G.SigmaEstimation(self.windows, self.pred_len, self.dataset_nf, 512,
                                                       self.rolling_length, native=self._native)

    def _prior_variance(self, ctx, series, n_windows, first, step, scale_input):
        if self.cond_pred_model_g is None:                         # 'Guassian': gx = 1  (:457-458)
            return torch.ones((n_windows, series.shape[0], self.pred_len, self.dataset_nf), device=ctx.device)
        mean = self.scaler_mean if scale_input else None
        std = self.scaler_std if scale_input else None
        return ctx.sigma_head(series, n_windows, first, step, mean, std, add_eps=self.EPS)     # + EPS (:450)

    def evaluation_step(self, batch, noise: Optional[torch.Tensor] = None, y0hat: Optional[torch.Tensor] = None):
        if self.needs_backbone and y0hat is None and self.cond_pred_model is None:
            raise NotImplementedError(_F_BACKBONE_MSG.format(self.train_model_select))
        return self._evaluation_step(batch, y0hat, noise)


class NsDiff_model(_NsDiffBase):
    """Plain NsDiff (:17-268): f(x) (csrc/nst.hip), g(x), sampler and reduction all native.  'pretrain_g' is the g-only
    configuration of the shipped ews_results/NsDiff_preg checkpoint, 'pretrain_f' the f-only one."""

    def __init__(self, net_param, train_model_select, pretrain_f_path="results/pre_model_F",
                 pretrain_g_path="results/pre_model_G"):
        super().__init__()
        self._init_common(net_param)
        self.train_model_select = train_model_select
        self.load_pretrain = net_param.get("load_pretrain", False)
        self.freeze_pretrain = net_param.get("freeze_pretrain", False)
        self.cond_pred_model = None
        if train_model_select == "pretrain_g":
            self.cond_pred_model_g = G.SigmaEstimation(self.windows, self.pred_len, self.dataset_nf, 512,
                                                       self.rolling_length, native=self._native)
        elif train_model_select == "NsDiff_model":
            self.needs_backbone = True
            self.model = NsDiff_net(self.configs, self.device)
            self.cond_pred_model = self._build_backbone(net_param)
            self.cond_pred_model_g = G.SigmaEstimation(self.windows, self.pred_len, self.dataset_nf, 512,
                                                       self.rolling_length, native=self._native)
        elif train_model_select == "pretrain_f":                   # f(x) alone (reference :88-91): no sampler to run
            self.needs_backbone = True
            self.cond_pred_model = self._build_backbone(net_param)
            if self.cond_pred_model is None:
                raise NotImplementedError(_F_BACKBONE_MSG.format(train_model_select))
        else:
            raise ValueError("train_model_select should be in ['NsDiff_model', 'pretrain_f', 'pretrain_g']")

    def _prior_variance(self, ctx, series, n_windows, first, step, scale_input):
        mean = self.scaler_mean if scale_input else None
        std = self.scaler_std if scale_input else None
        return ctx.sigma_head(series, n_windows, first, step, mean, std, add_eps=0.0)           # no EPS (:223)

    def evaluation_step(self, batch, y0hat: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        if getattr(self, "model", None) is None:
            raise RuntimeError("train_model_select={!r} has no diffusion network to sample from".format(self.train_model_select))
        if y0hat is None and self.cond_pred_model is None:
            raise NotImplementedError(_F_BACKBONE_MSG.format(self.train_model_select))
        return self._evaluation_step(batch, y0hat, noise)


SPATIAL_CONFIG_KEYS = ns_Transformer.CONFIG_KEYS + ("fT_h", "spatial_layers")

_F_SPATIAL_MSG = (
    "NsDiff_spatial's mean backbone f(x) is the reference's `Model_spatial` (NsDiff/mu_backbone.py:186-340: the ns-Transformer with graph "
    "convolutions between encoder and decoder), and net_param lacks its hyper-parameters (" + ", ".join(SPATIAL_CONFIG_KEYS) + "), so none was "
    "built.  Add them, pass a precomputed y_0_hat [V, pred_len, F] (`y0hat=`), or attach your own module with the reference's call signature "
    "`(batch_x, dec_inp, edge_index) -> (y_0_hat, _)` via `attach_backbone(module)`.")


class NsDiff_model_spatial(_NsDiffBase):
    """NsDiff over a graph (reference NsDiff_model.py:497-790): g(x) by the sigma-head kernel, the graph U-Net eps / sigma network and the
    NsDiff reverse loop by csrc/stg.hip (engine mode 1) behind mpv_nss_*; f(x) = `Model_spatial` by csrc/nst.hip's spatial mode when net_param
    carries its hyper-parameters (d_model, n_heads, d_ff, e_layers, d_layers, activation, p_hidden_dims, p_hidden_layers, fT_h, spatial_layers),
    else an input (`y0hat=` / attach_backbone).

    `evaluation_step(gdata)` takes the reference's graph batch (`.x` [V, >= windows, F], `.edge_index` [2, E], `.num_nodes`) and returns
    (outs [V, pred_len, F, n_z_samples] on the CPU, batch_y | None).  Two reference behaviours are kept as they are: rows are tiled
    node-major (row v*P + p, :756-763) while the duplicated edge index is copy-major (copy r // V, :783-790), and with parallel_sample = 1
    the reference never defines the edge index it passes on (UnboundLocalError, :751-753) -- here that case uses the plain edge index."""

    batched_history = True          # the harness hands `sample_windows` all windows of a launch as [W, V, history, F] (+ the graph)

    def __init__(self, net_param, train_model_select, pretrain_f_path="results/pre_model_F", pretrain_g_path="results/pre_model_G"):
        super().__init__()
        self._init_common(net_param)
        self.train_model_select = train_model_select
        self.load_pretrain = net_param.get("load_pretrain", False)
        self.freeze_pretrain = net_param.get("freeze_pretrain", False)
        self.cond_pred_model = None
        if train_model_select == "NsDiff_model":
            self.needs_backbone = True
            self.model = NsDiff_net_spatial(self.configs, self.device)
            self.cond_pred_model = self._build_spatial_backbone(net_param)                  # reference :539-548
            self.cond_pred_model_g = G.SigmaEstimation(self.windows, self.pred_len, self.dataset_nf, 512, self.rolling_length, native=self._native)
        elif train_model_select == "pretrain_g":
            self.cond_pred_model_g = G.SigmaEstimation(self.windows, self.pred_len, self.dataset_nf, 512, self.rolling_length, native=self._native)
        elif train_model_select == "pretrain_f":                                            # reference :561: f(x) alone
            self.needs_backbone = True
            self.cond_pred_model = self._build_spatial_backbone(net_param)
            if self.cond_pred_model is None:
                raise NotImplementedError(_F_SPATIAL_MSG)
        else:
            raise ValueError("train_model_select should be in ['NsDiff_model', 'pretrain_f', 'pretrain_g']")

    def scaler_fit(self, data):                                     # reference :588-593: statistics over nodes AND time
        std = data.std(axis=(0, 1))
        std[std == 0] = 1
        self.scaler_mean = data.mean(axis=(0, 1))
        self.scaler_std = std

    def _build_spatial_backbone(self, net_param):
        if all(k in net_param for k in SPATIAL_CONFIG_KEYS):
            return ns_Transformer.Model_spatial(self.configs, native=self._native)
        return None

    def _native_backbone(self) -> bool:
        return isinstance(getattr(self, "cond_pred_model", None), ns_Transformer.Model_spatial)

    def _upload_native(self, ctx):
        if getattr(self, "model", None) is not None:
            ctx.load_nss(self.model.diffussion_model, self.model.coef_rows, self.diffusion_steps, gemm=self.gemm_mode if self.gemm_mode in ("f16x3", "f32") else "f16x3")
        g = getattr(self, "cond_pred_model_g", None)
        if g is not None:
            ctx.load_sigma(g.seq_len, g.kernel_size, g.pred_len, g.enc_in, g.hidden_size, g.native_weights())
        if self._native_backbone():
            ctx.load_nst(self.cond_pred_model)

    @staticmethod
    def duplicate_edge_index(parallel_sampling, edge_index, num_nodes, device):
        """Reference :783-790 (kept for callers; the library stores ONE graph and offsets every copy)."""
        edge_index = edge_index.reshape((2, 1, -1))
        indent = torch.arange(0, parallel_sampling).view(1, -1, 1).to(device) * num_nodes
        return (edge_index + indent).reshape((2, -1))

    def _spatial_mean(self, batch_x, edge_index):
        """y_0_hat [V, pred_len, F] of one graph's scaled windows [V, windows, F] (reference :744)."""
        if self.cond_pred_model is None:
            raise NotImplementedError(_F_SPATIAL_MSG)
        if self._native_backbone():
            return self.cond_pred_model(batch_x, None, edge_index)[0]
        dev = next(self.cond_pred_model.parameters(), torch.zeros(0, device=self.scaler_mean.device)).device
        with torch.no_grad():
            y0hat, _ = self.cond_pred_model(batch_x.to(dev), self._decoder_input(batch_x.to(dev)), edge_index.to(dev))
        return y0hat

    def spatial_mean_windows(self, history: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
        """f(x) of W windows of one graph in ONE call: history [W, V, windows, F] (scaled) -> y_0_hat [W, V, pred_len, F] on device.  The native
        backbone gathers windows out of a series, so the windows are laid end to end per node (step = windows); an attached module is called
        window by window with the reference's signature."""
        if self.cond_pred_model is None:
            raise NotImplementedError(_F_SPATIAL_MSG)
        W, V = history.shape[0], history.shape[1]
        if not self._native_backbone():
            return torch.stack([self._spatial_mean(history[w], edge_index).to(history.device) for w in range(W)])
        ctx = self._native.context()
        ctx.nst_set_graph(edge_index, V)
        series = history.to(device=ctx.device, dtype=torch.float32).permute(1, 0, 2, 3).reshape(V, W * self.windows, self.dataset_nf).contiguous()
        return ctx.nst_forward(series, W, 0, self.windows, gemm=self.cond_pred_model.gemm_mode)

    def sample_windows(self, history: torch.Tensor, edge_index: torch.Tensor, y0hat: Optional[torch.Tensor] = None, *, noise: Optional[torch.Tensor] = None,
                       seed: Optional[int] = None) -> torch.Tensor:
        """history [W, V, windows, F] (scaled), the ONE graph's edge_index [2, E], y0hat [W, V, pred_len, F] (None: f(x) = the native Model_spatial
        or the attached module) -> samples [W, V, pred_len, F, n_z_samples] on device: one sigma-head launch over all W*V rows, then every
        window's parallel_sample graph copies side by side through the T-step loop, n_z_samples / parallel_sample passes."""
        if getattr(self, "model", None) is None:
            raise RuntimeError("train_model_select={!r} has no diffusion network to sample from".format(self.train_model_select))
        S, P = self._check_sampling_config()
        ctx = self._native.context()
        hist = history.to(device=ctx.device, dtype=torch.float32).contiguous()
        if hist.dim() != 4 or hist.shape[2] != self.windows or hist.shape[3] != self.dataset_nf:
            raise ValueError("history must be [W, V, {}, {}], got {}".format(self.windows, self.dataset_nf, tuple(hist.shape)))
        W, V = hist.shape[0], hist.shape[1]
        gx = ctx.sigma_head(hist.reshape(W * V, self.windows, self.dataset_nf), 1, 0, 1)[0].reshape(W, V, self.pred_len, self.dataset_nf).contiguous()   # no EPS (:748)
        if y0hat is None:
            y0hat = self.spatial_mean_windows(hist, edge_index)
        y0 = y0hat.to(device=ctx.device, dtype=torch.float32).reshape(W, V, self.pred_len, self.dataset_nf).contiguous()
        if noise is None and self.noise_source == "torch_cpu":
            noise = self.host_noise(W, V)
        if noise is not None:
            noise = noise.to(device=ctx.device, dtype=torch.float32).contiguous()
        elif seed is None:
            seed = _draw_seed()
        return ctx.nss_sample(y0, gx, edge_index, S, P, noise=noise, seed=seed or 0)

    def evaluation_step(self, gdata, y0hat: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        batch = gdata.x
        edge_index = torch.as_tensor(gdata.edge_index).reshape(2, -1)
        if batch.dim() != 3 or batch.shape[1] < self.windows:
            raise ValueError("gdata.x must be [V, >= windows ({}), F], got {}".format(self.windows, tuple(batch.shape)))
        if int(gdata.num_nodes) != batch.shape[0]:
            raise ValueError("gdata.num_nodes ({}) does not match gdata.x ({} rows)".format(gdata.num_nodes, batch.shape[0]))
        batch_y = None
        if batch.shape[1] - self.windows >= self.pred_len:
            batch_y = batch[:, self.windows:, :].to(self.device)
            assert batch_y.size(1) == self.pred_len, "pred_len is not equal to the length of the prediction"
        x = batch[:, : self.windows, :]
        if y0hat is None:
            y0hat = self._spatial_mean(x, edge_index)
        if noise is not None and noise.dim() != 6:
            noise = noise.reshape(1, -1, self.diffusion_steps, x.shape[0] * int(self.configs.parallel_sample), self.pred_len, self.dataset_nf)
        samples = self.sample_windows(x.unsqueeze(0), edge_index, y0hat.unsqueeze(0), noise=noise)
        outs = samples[0].cpu()                                   # [V, O, F, S], the reference's `outs`
        assert tuple(outs.shape[1:]) == (self.pred_len, self.dataset_nf, int(self.configs.n_z_samples))
        return outs, batch_y
