"""DiffusionTS model object for the MPV path (mirrors reference DiffusionTS/DiffusionTS_model.py:9-109).

Same constructor (`DiffusionTS_model(net_param)`), attribute and parameter names (`scaler`, `scaler_mean`, `scaler_std`,
`gt_mask`, `model` = Diffusion_TS holding `model` = Transformer), defaults, scaler methods, the ValueError on
`n_z_samples % parallel_sample`, and `evaluation_step(batch) -> (outs [B, pred_len, F, n_z_samples] on CPU, batch_y | None)`.

The computation -- x0-predicting transformer forward, its input gradient for the Langevin infill, the DDIM loop -- runs in the
HIP library (mpv_dts_sample); `sample_windows` batches many sliding windows into one call.  The reference tiles the batch
SAMPLE-major (`x.repeat(par, 1, 1)`, :92) but un-tiles it as `(B, par, ...)` (:102-104), so row v of `outs` mixes samples of
different nodes whenever B > 1; MPV averages over all rows, so the quirk changes the number and is reproduced as is.

Noise: `noise_source = "philox"` (default; device draws) or `"torch_cpu"` (every `torch.randn` of the reference drawn on the
host in the reference's order and uploaded: reproduces a reference CPU run under the same `torch.manual_seed`).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from ..NsDiff.native_handle import NativeHandle
from ..NsDiff.NsDiff_model import _draw_seed
from .DiffusionTS import Diffusion_TS


class DiffusionTS_model(nn.Module):
    batched_history = True          # the harness hands `sample_windows` all windows of a launch as [W, V, history, F]

    def __init__(self, net_param):
        super().__init__()
        self.device = net_param["device"]
        self.dataset_nf = net_param["dataset_nf"]
        self.windows = net_param["windows"]
        self.pred_len = net_param["pred_len"]
        self.seq_len = net_param["seq_len"] = self.windows
        self.label_len = net_param["label_len"] = self.windows // 2
        self.n_z_samples = net_param.get("n_z_samples", 100)
        self.parallel_sample = net_param.get("parallel_sample", min(10, self.n_z_samples))
        self.sampling_timesteps = net_param.get("diffusion_steps", 100)
        self.scaler = net_param.get("scaler_type", None)
        self.noise_source = net_param.get("noise_source", "philox")
        if self.noise_source not in ("philox", "torch_cpu"):
            raise ValueError("noise_source must be 'philox' or 'torch_cpu'")
        self.configs = SimpleNamespace(**net_param)
        self.register_buffer("scaler_mean", torch.zeros(self.dataset_nf))
        self.register_buffer("scaler_std", torch.ones(self.dataset_nf))
        self.model = Diffusion_TS(
            seq_length=self.windows + self.pred_len, feature_size=self.dataset_nf, n_layer_enc=net_param.get("n_layer_enc", 3),
            n_layer_dec=net_param.get("n_layer_dec", 6), d_model=net_param.get("d_model", 64), timesteps=net_param.get("timesteps", 100),
            sampling_timesteps=self.sampling_timesteps, loss_type=net_param.get("loss_type", "l2"),
            beta_schedule=net_param.get("beta_schedule", "cosine"), n_heads=net_param.get("n_heads", 4),
            mlp_hidden_times=net_param.get("mlp_hidden_times", 4), eta=net_param.get("eta", 0.0), attn_pd=net_param.get("attn_pd", 0.0),
            resid_pd=net_param.get("resid_pd", 0.0), kernel_size=net_param.get("kernel_size"), padding_size=net_param.get("padding_size"),
            use_ff=net_param.get("use_ff", True), reg_weight=net_param.get("reg_weight"))
        gt_mask = torch.cat([torch.ones(self.windows, self.dataset_nf, dtype=torch.bool),
                             torch.zeros(self.pred_len, self.dataset_nf, dtype=torch.bool)], dim=0)
        self.register_buffer("gt_mask", gt_mask)
        object.__setattr__(self, "_native", NativeHandle())
        self._native.bind(self)

    # ---- scaler (:62-70) --------------------------------------------------------------------------------------------
    def scaler_fit(self, data):
        data_std = data.std(axis=0)
        data_std[data_std == 0] = 1
        self.scaler_mean = data.mean(axis=0)
        self.scaler_std = data_std

    def scaler_transform(self, data):
        return (data - self.scaler_mean) / self.scaler_std

    def scaler_inverse_transform(self, data):
        return (data * self.scaler_std) + self.scaler_mean

    # ---- weights -> device ------------------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, *args, **kwargs):
        out = super().load_state_dict(state_dict, *args, **kwargs)
        self._native.mark_dirty()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._native.mark_dirty()
        return out

    def refresh_native(self):
        self._native.mark_dirty()

    def _upload_native(self, ctx):
        tf = self.model.model
        L = self.windows + self.pred_len
        lin_space = torch.arange(1, L + 1, 1) / (L + 1)                                    # TrendBlock.poly_space, transformer.py:26-27
        poly = torch.stack([lin_space ** float(p + 1) for p in range(3)], dim=0)
        ctx.load_dts(L, self.dataset_nf, tf.n_embd, tf.n_heads, tf.n_layer_enc, tf.n_layer_dec, tf.mlp_hidden_times,
                     self.model.num_timesteps, tf.combine_taps, tf.native_weight_list(), tf.ada_tables(self.model.num_timesteps),
                     torch.fft.rfftfreq(L), poly, gemm=self.configs.__dict__.get("gemm_mode", "f16x3"))

    def native_context(self):
        return self._native.context()

    # ---- sampling ---------------------------------------------------------------------------------------------------
    def _chunking(self):
        par = min(int(self.parallel_sample), int(self.n_z_samples))
        if self.n_z_samples % par != 0:
            raise ValueError("n_z_samples must be divisible by parallel_sample")            # :83-84
        return int(self.n_z_samples), par

    def _plan(self) -> torch.Tensor:
        lr = self.configs.__dict__.get("infill_learning_rate", 5e-2)
        plan = self.model.infill_plan(self.sampling_timesteps, lr)
        t = plan[:, 0].long()
        extra = torch.stack([self.model.sqrt_alphas_cumprod.cpu()[t], self.model.sqrt_one_minus_alphas_cumprod.cpu()[t]], dim=1)
        return torch.cat([plan[:, :8], extra, plan[:, 8:9]], dim=1).contiguous()        # MPV_DTS_PLAN_COLS = 11: sigma last

    def host_noise(self, n_windows: int, nodes: int) -> torch.Tensor:
        """Every normal draw of `n_windows` evaluation_step calls with torch.randn on the CPU in the reference's order (window,
        chunk, then SURVEY A.3), re-laid as mpv_dts_sample wants it: [chunks][draws][W][par*nodes][L][F]."""
        S, par = self._chunking()
        plan = self._plan()
        draws = 1 + sum(2 + int(r[6]) for r in plan.tolist() if r[1] >= 0)
        shape = (par * nodes, self.windows + self.pred_len, self.dataset_nf)
        out = torch.empty((S // par, draws, n_windows) + shape)
        for w in range(n_windows):
            for c in range(S // par):
                for k in range(draws):
                    out[c, k, w] = torch.randn(shape)
        return out

    def sample_windows(self, history: torch.Tensor, *, noise: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> torch.Tensor:
        """history [W, V, windows, F] (scaled) -> samples [W, V, pred_len, F, n_z_samples] on device, all windows in one call."""
        S, par = self._chunking()
        ctx = self._native.context()
        hist = history.to(device=ctx.device, dtype=torch.float32).contiguous()
        if hist.dim() != 4 or hist.shape[2] != self.windows or hist.shape[3] != self.dataset_nf:
            raise ValueError("history must be [W, V, {}, {}], got {}".format(self.windows, self.dataset_nf, tuple(hist.shape)))
        if noise is None and self.noise_source == "torch_cpu":
            noise = self.host_noise(hist.shape[0], hist.shape[1])
        if noise is not None:
            noise = noise.to(device=ctx.device, dtype=torch.float32).contiguous()
        elif seed is None:
            seed = _draw_seed()
        coef = self.configs.__dict__.get("infill_coef", 1e-1)
        return ctx.dts_sample(hist, self._plan(), S, par, coef, noise=noise, seed=seed or 0)

    def evaluation_step(self, batch, noise: Optional[torch.Tensor] = None):
        batch_x = batch[:, : self.windows, :]
        if batch.shape[1] - self.windows >= self.pred_len:
            batch_y = batch[:, self.windows: self.windows + self.pred_len, :].to(self.device)
        else:
            batch_y = None
        S, par = self._chunking()
        if noise is not None and noise.dim() != 6:                                        # a flat record of the reference's draws
            L = self.windows + self.pred_len
            noise = noise.reshape(S // par, -1, 1, par * batch_x.shape[0], L, self.dataset_nf)
        samples = self.sample_windows(batch_x.unsqueeze(0), noise=noise)
        return samples[0].cpu(), batch_y
