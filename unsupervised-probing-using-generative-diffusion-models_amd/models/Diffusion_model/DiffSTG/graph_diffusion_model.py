"""DiffSTG model object for the MPV path (mirrors reference DiffSTG/graph_diffusion_model.py:13-282).

Same constructor (`DiffSTG(net_param)`), attribute / parameter names (`model` = UGnet, `scaler`, `scaler_mean`, `scaler_std`,
`diffusion`, `T_h`, `T_p`, ...), scaler methods and
`evaluation_step(data) -> (splitted_predict_x0 [V, T_h+T_p, 1, parallel*sequential] on CPU, x0_truth | None)` where `data` is any
object with `.x [V, >= T_h, F]`, `.edge_index [2, E]` (source row 0, target row 1, both directions listed, as
torch_geometric's `from_networkx` gives it) and `.num_nodes`.

The U-Net (causal TCN blocks, LayerNorm, the k = T+1 down/up-sampling convolutions as MFMA GEMMs, the gated graph convolution as
a CSR gather over the `parallel_sampling`-duplicated graph) and the DDIM / DDPM posterior run in the HIP library (mpv_stg_*).
The posterior's coefficients are float64 NumPy expressions applied as Python floats, as in the reference (:51-69).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from ..NsDiff.native_handle import NativeHandle
from ..NsDiff.NsDiff_model import _draw_seed
from .diffusion_schedulers import GaussianDiffusion, posterior_plan
from .ugnet import UGnet


class graph_Diffusion_model(nn.Module):
    def __init__(self, net_param):
        super().__init__()
        self.diffusion_steps = net_param["diffusion_steps"]
        self.inference_diffusion_steps = net_param["inference_diffusion_steps"]
        self.inference_trick = net_param["inference_trick"] if net_param.get("inference_trick") else "ddim"
        self.device = net_param["device"]
        self.diffusion_schedule = net_param["diffusion_schedule"]
        self.inference_schedule = net_param["inference_schedule"]
        self.loss_weight_schedule = net_param["loss_weight_schedule"]
        self.parallel_sampling = net_param["parallel_sampling"]
        self.sequential_sampling = net_param["sequential_sampling"]
        self.sparse = True
        self.diffusion = GaussianDiffusion(T=self.diffusion_steps, schedule=self.diffusion_schedule,
                                           loss_weight_schedule=self.loss_weight_schedule)


class DiffSTG(graph_Diffusion_model):
    batched_history = True          # the harness hands `sample_windows` all windows of a launch as [W, V, history, F]

    def __init__(self, net_param):
        super().__init__(net_param=net_param)
        self.T_p, self.T_h, self.F = net_param["T_p"], net_param["T_h"], net_param["F"]
        self.T = self.T_p + self.T_h
        self.mask_ratio = net_param["mask_ratio"]
        self.model = UGnet(net_param=net_param)
        self.scaler = net_param["scaler_type"]
        self.noise_source = net_param.get("noise_source", "philox")
        if self.noise_source not in ("philox", "torch_cpu"):
            raise ValueError("noise_source must be 'philox' or 'torch_cpu'")
        self.register_buffer("scaler_mean", torch.zeros(self.F))
        self.register_buffer("scaler_std", torch.zeros(self.F))
        self.pred_len, self.dataset_nf = self.T_p, self.F
        object.__setattr__(self, "_native", NativeHandle())
        self._native.bind(self)

    def scaler_fit(self, data):
        data_std = data.std(axis=(0, 1))
        data_std[data_std == 0] = 1
        self.scaler_mean = data.mean(axis=(0, 1))
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
        ctx.load_stg(self.model)

    def native_context(self):
        return self._native.context()

    # ---- sampling ---------------------------------------------------------------------------------------------------
    def _plan(self) -> torch.Tensor:
        return torch.from_numpy(posterior_plan(self.diffusion, self.inference_schedule, self.inference_diffusion_steps,
                                               self.inference_trick))

    def host_noise(self, n_windows: int, nodes: int) -> torch.Tensor:
        """Every normal draw of `n_windows` evaluation_step calls with torch.randn on the CPU in the reference's order (per
        sequential pass: x_T, then one z per DDPM step), laid out as mpv_stg_sample wants it: [S][draws][W][P*V][T][F]."""
        plan = self._plan()
        draws = 1 + int((plan[:, 1] > 0).sum())
        shape = (self.parallel_sampling * nodes, self.T, self.F)
        out = torch.empty((self.sequential_sampling, draws, n_windows) + shape)
        for w in range(n_windows):
            for s in range(self.sequential_sampling):
                for k in range(draws):
                    out[s, k, w] = torch.randn(shape)
        return out

    def sample_windows(self, history: torch.Tensor, edge_index: torch.Tensor, *, noise: Optional[torch.Tensor] = None,
                       seed: Optional[int] = None) -> torch.Tensor:
        """history [W, V, T_h, F] (scaled), edge_index [2, E] of ONE graph -> samples [W, V, T, F, P*S] on device; the W windows
        and the P parallel samples run as W*P disjoint copies of the graph in one batch."""
        ctx = self._native.context()
        hist = history.to(device=ctx.device, dtype=torch.float32).contiguous()
        if hist.dim() != 4 or hist.shape[2] != self.T_h or hist.shape[3] != self.F:
            raise ValueError("history must be [W, V, {}, {}], got {}".format(self.T_h, self.F, tuple(hist.shape)))
        if noise is None and self.noise_source == "torch_cpu":
            noise = self.host_noise(hist.shape[0], hist.shape[1])
        if noise is not None:
            noise = noise.to(device=ctx.device, dtype=torch.float32).contiguous()
        elif seed is None:
            seed = _draw_seed()
        return ctx.stg_sample(hist, edge_index, self._plan(), self.parallel_sampling, self.sequential_sampling, noise=noise, seed=seed or 0)

    def evaluation_step(self, data, noise: Optional[torch.Tensor] = None):
        x = data.x
        history = x[:, : self.T_h, :]
        x0_truth = None
        if x.shape[1] - self.T_h >= self.T_p:
            future = x[:, self.T_h:, :]
            assert future.size(1) == self.T_p, "pred_len is not equal to the length of the prediction"
            x0_truth = torch.cat([history, future], dim=1).to(self.device)
        if noise is not None and noise.dim() != 6:                                        # a flat record of the reference's draws
            noise = noise.reshape(self.sequential_sampling, -1, 1, self.parallel_sampling * history.shape[0], self.T, self.F)
        samples = self.sample_windows(history.unsqueeze(0), data.edge_index, noise=noise)   # [1, V, T, F, P*S]
        if self.F != 1:
            raise ValueError("DiffSTG.evaluation_step reshapes its samples with a feature axis of 1 (reference :277)")
        return samples[0].cpu(), x0_truth
