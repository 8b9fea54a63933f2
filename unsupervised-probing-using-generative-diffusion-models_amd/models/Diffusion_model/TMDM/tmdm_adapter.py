"""TMDM model object for the MPV path (mirrors reference TMDM/tmdm_adapter.py:25-155).

Same constructor (`TMDM_model(net_param)`), attribute names (`scaler`, `scaler_mean`, `scaler_std`, `model`,
`cond_pred_model`, `configs`, `label_len`, ...), defaults, scaler methods and
`evaluation_step(batch) -> (outs [B, pred_len, F, n_z_samples] on CPU, batch_y | None)`.

The sampler (epsilon MLP + CARD reverse loop + tile/un-tile, :122-155) runs in the fused HIP chain kernel.
The prior `y_0_hat` (:122-123) is the mean head of the reference's ns-Transformer VAE: `cond_pred_model` is the holder of
tmdm_ns_transformer.py, run by csrc/nst.hip (its third-party `torch_timeseries` layers are restated -- SURVEY.md section 8c: parity
unpinned).  `y_0_hat=` on `evaluation_step` overrides it, and any module with the reference's call signature
`(batch_x, None, dec_inp, None) -> (_, y_0_hat, _, _)` may be assigned to `cond_pred_model` instead.

Because the sampler is pointwise along the sequence, only the last `pred_len` of the `label_len + pred_len`
positions (the ones the reference keeps, :152) are sampled.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
import torch.nn as nn

from ..NsDiff.native_handle import NativeHandle
from ..NsDiff.NsDiff_model import _draw_seed
from . import tmdm_ns_transformer as ns_Transformer
from .TMDM import TMDM

_PRIOR_MSG = ("TMDM's prior network was removed from this model (cond_pred_model is None).  Pass y_0_hat=[B, label_len+pred_len, F] "
              "(or [B, pred_len, F]) to evaluation_step, or assign a module to `cond_pred_model`.")


class TMDM_model(nn.Module):
    def __init__(self, net_param):
        super().__init__()
        self.device = net_param["device"]
        self.dataset_nf = net_param["dataset_nf"]
        self.windows = net_param["windows"]
        self.pred_len = net_param["pred_len"]
        self.seq_len = net_param["seq_len"] = self.windows
        self.label_len = net_param["label_len"] = net_param.get("label_len", self.windows // 2)
        self.diffusion_steps = net_param.get("diffusion_steps", 100)
        self.n_z_samples = net_param.get("n_z_samples", 100)
        self.parallel_sample = net_param.get("parallel_sample", min(10, self.n_z_samples))
        self.scaler = net_param.get("scaler_type", None)
        self.k_z = net_param.get("k_z", 0.01)
        self.gemm_mode = net_param.get("gemm_mode", "f16x3")         # "f16x3" (split-f16 matrix cores, fp32-equivalent) | "f32" (exact)
        self.noise_source = net_param.get("noise_source", "philox")
        if self.noise_source not in ("philox", "torch_cpu"):
            raise ValueError("noise_source must be 'philox' or 'torch_cpu'")
        net_param.setdefault("enc_in", self.dataset_nf)
        net_param.setdefault("timesteps", self.diffusion_steps)
        for key, default in (("dec_in", self.dataset_nf), ("c_out", self.dataset_nf), ("output_attention", False), ("d_model", 64), ("n_heads", 4),
                             ("d_ff", 128), ("activation", "gelu"), ("e_layers", 2), ("d_layers", 1), ("p_hidden_dims", [64, 64]),
                             ("p_hidden_layers", 2)):                          # the prior's defaults, reference :38-55
            net_param.setdefault(key, default)
        self.configs = SimpleNamespace(**net_param)
        self.register_buffer("scaler_mean", torch.zeros(self.dataset_nf))
        self.register_buffer("scaler_std", torch.ones(self.dataset_nf))
        self.model = TMDM(self.configs, self.device)
        object.__setattr__(self, "_native", NativeHandle())
        self._native.bind(self)
        self.cond_pred_model = ns_Transformer.Model(self.configs, native=self._native)        # reference :70

    # ---- scaler (:73-83) ------------------------------------------------------------
    def scaler_fit(self, data):
        data_std = data.std(axis=0)
        data_std[data_std == 0] = 1
        self.scaler_mean = data.mean(axis=0)
        self.scaler_std = data_std

    def scaler_transform(self, data):
        return (data - self.scaler_mean) / self.scaler_std

    def scaler_inverse_transform(self, data):
        return (data * self.scaler_std) + self.scaler_mean

    def _decoder_input(self, batch_x):
        dec_inp_pred = torch.zeros(batch_x.size(0), self.pred_len, self.dataset_nf, device=batch_x.device)
        return torch.cat([batch_x[:, -self.label_len:, :], dec_inp_pred], dim=1)

    # ---- weights ----------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True, **kwargs):
        """Reference checkpoints also hold `cond_pred_model.*` and `model.enc_embedding.*`; neither is on
        the native path (see module docstring), so those entries are dropped before a strict load."""
        kept = {k: v for k, v in state_dict.items()
                if not (k.startswith("model.enc_embedding.") or (self.cond_pred_model is None and k.startswith("cond_pred_model.")))}
        out = super().load_state_dict(kept, strict=strict, **kwargs)
        self._native.mark_dirty()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._native.mark_dirty()
        return out

    def refresh_native(self):
        self._native.mark_dirty()

    def _upload_native(self, ctx):
        w = self.model.native_weights()
        ctx.load_tmdm(self.model.num_timesteps, self.dataset_nf, w["lin_w"], w["lin_b"], w["embed"], w["eps_w"],
                      w["eps_b"], w["coef"])
        if isinstance(self.cond_pred_model, ns_Transformer.Model):
            ctx.load_nst(self.cond_pred_model)

    def native_context(self):
        return self._native.context()

    # ---- sampling ---------------------------------------------------------------------
    def _chunking(self):
        par = min(int(self.parallel_sample), int(self.n_z_samples))
        if par < 1 or self.n_z_samples % par != 0:
            raise ValueError("n_z_samples must be divisible by parallel_sample")          # :126-127
        return int(self.n_z_samples), par

    def host_noise(self, n_windows: int, nodes: int) -> torch.Tensor:
        """The reference's normals (torch.randn on the CPU, its order: window, chunk, z_T then one z per step;
        each draw [nodes*P, label_len+pred_len, F]), cut down to the pred_len positions that are sampled."""
        S, P = self._chunking()
        L, O, T = self.label_len + self.pred_len, self.pred_len, self.model.num_timesteps
        out = torch.empty((n_windows, S // P, T, nodes * P, O, self.dataset_nf))
        for w in range(n_windows):
            for c in range(S // P):
                for k in range(T):
                    out[w, c, k] = torch.randn(nodes * P, L, self.dataset_nf)[:, L - O:, :]
        return out

    def sample_windows(self, y_0_hat: torch.Tensor, *, noise: Optional[torch.Tensor] = None,
                       seed: Optional[int] = None, offset: int = 0) -> torch.Tensor:
        """y_0_hat [W, B, pred_len, F] (prior mean at the kept positions) -> samples [W, B, pred_len, F, S]
        on device, one launch for all windows."""
        S, P = self._chunking()
        ctx = self._native.context()
        prior = y_0_hat.to(device=ctx.device, dtype=torch.float32).contiguous()
        if prior.dim() != 4 or prior.shape[2] != self.pred_len or prior.shape[3] != self.dataset_nf:
            raise ValueError("y_0_hat must be [W, B, {}, {}], got {}".format(self.pred_len, self.dataset_nf, tuple(prior.shape)))
        if noise is None and self.noise_source == "torch_cpu":
            noise = self.host_noise(prior.shape[0], prior.shape[1])
        if noise is not None:
            noise = noise.to(device=ctx.device, dtype=torch.float32).contiguous()
        elif seed is None:
            seed = _draw_seed()
        return ctx.sample_tmdm(prior, S, P, noise=noise, seed=seed or 0, offset=offset, gemm=self.gemm_mode)

    def evaluation_step(self, batch, y_0_hat: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None):
        batch_x = batch[:, : self.windows, :]
        if batch.shape[1] - self.windows >= self.pred_len:
            batch_y = batch[:, self.windows: self.windows + self.pred_len, :].to(self.device)
        else:
            batch_y = None
        self._chunking()
        if y_0_hat is None:
            if self.cond_pred_model is None:
                raise NotImplementedError(_PRIOR_MSG)
            with torch.no_grad():
                xb = batch_x.to(self.device)
                _, y_0_hat, _, _ = self.cond_pred_model(xb, None, self._decoder_input(xb), None)
        full = self.label_len + self.pred_len
        if y_0_hat.dim() != 3 or y_0_hat.shape[0] != batch_x.shape[0] or y_0_hat.shape[1] not in (full, self.pred_len):
            raise ValueError("y_0_hat must be [B, {} or {}, F], got {}".format(full, self.pred_len, tuple(y_0_hat.shape)))
        prior = y_0_hat[:, -self.pred_len:, :]
        samples = self.sample_windows(prior.unsqueeze(0), noise=noise)
        return samples[0].cpu(), batch_y
