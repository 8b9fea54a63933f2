"""Diffusion_TS holder (mirrors reference DiffusionTS/DiffusionTS.py:34-124 in buffer / module names) plus the scalar
schedules of its DDIM + Langevin-infill sampler (:277-310, :359-381) that the host hands to mpv_dts_sample."""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from .diffusionts_transformer import Transformer


def cosine_beta_schedule(timesteps, s=0.008):                       # reference :22-32 (float64)
    steps = timesteps + 1
    x = torch.linspace(0, timesteps, steps, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def linear_beta_schedule(timesteps):                                # reference :15-19
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


class Diffusion_TS(nn.Module):
    def __init__(self, seq_length, feature_size, n_layer_enc=3, n_layer_dec=6, d_model=None, timesteps=1000,
                 sampling_timesteps=None, loss_type="l1", beta_schedule="cosine", n_heads=4, mlp_hidden_times=4, eta=0.0,
                 attn_pd=0.0, resid_pd=0.0, kernel_size=None, padding_size=None, use_ff=True, reg_weight=None, **kwargs):
        super().__init__()
        self.eta, self.use_ff = eta, use_ff
        self.seq_length, self.feature_size = seq_length, feature_size
        self.model = Transformer(n_feat=feature_size, n_channel=seq_length, n_layer_enc=n_layer_enc, n_layer_dec=n_layer_dec,
                                 n_embd=d_model, n_heads=n_heads, mlp_hidden_times=mlp_hidden_times,
                                 conv_params=[kernel_size, padding_size])
        if beta_schedule == "linear":
            betas = linear_beta_schedule(timesteps)
        elif beta_schedule == "cosine":
            betas = cosine_beta_schedule(timesteps)
        else:
            raise ValueError("unknown beta schedule {}".format(beta_schedule))
        alphas = 1.0 - betas
        ac = torch.cumprod(alphas, dim=0)
        ac_prev = torch.nn.functional.pad(ac[:-1], (1, 0), value=1.0)
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = sampling_timesteps if sampling_timesteps is not None else self.num_timesteps
        assert self.sampling_timesteps <= self.num_timesteps
        reg = lambda name, val: self.register_buffer(name, val.to(torch.float32))
        reg("betas", betas)
        reg("alphas_cumprod", ac)
        reg("alphas_cumprod_prev", ac_prev)
        reg("sqrt_alphas_cumprod", torch.sqrt(ac))
        reg("sqrt_one_minus_alphas_cumprod", torch.sqrt(1.0 - ac))
        reg("log_one_minus_alphas_cumprod", torch.log(1.0 - ac))
        reg("sqrt_recip_alphas_cumprod", torch.sqrt(1.0 / ac))
        reg("sqrt_recipm1_alphas_cumprod", torch.sqrt(1.0 / ac - 1))
        pv = betas * (1.0 - ac_prev) / (1.0 - ac)
        reg("posterior_variance", pv)
        reg("posterior_log_variance_clipped", torch.log(pv.clamp(min=1e-20)))
        reg("posterior_mean_coef1", betas * torch.sqrt(ac_prev) / (1.0 - ac))
        reg("posterior_mean_coef2", (1.0 - ac_prev) * torch.sqrt(alphas) / (1.0 - ac))
        reg("loss_weight", torch.sqrt(alphas) * torch.sqrt(1.0 - ac) / betas / 100)

    # ---- scalar plan of fast_sample_infill (reference :277-310, 359-381) ------------------------------------------
    def infill_plan(self, sampling_timesteps: int, learning_rate: float) -> torch.Tensor:
        """One row per DDIM pair (time, time_next), float32 [n_pairs][9]:
        0 time, 1 time_next, 2 sqrt_recip_alphas_cumprod[time], 3 sqrt_recipm1_alphas_cumprod[time],
        4 sqrt(alpha_next), 5 c = sqrt(1 - alpha_next - sigma^2), 6 K Langevin iterations, 7 their learning rate,
        8 sigma = eta sqrt((1 - alpha / alpha_next)(1 - alpha_next) / (1 - alpha))  (0 for eta = 0, the shipped configs);
        rows with time_next < 0 carry zeros in 4..8.  Evaluated with the reference's fp32 tensor ops (:294-297)."""
        T = self.num_timesteps
        times = list(reversed(torch.linspace(-1, T - 1, steps=sampling_timesteps + 1).int().tolist()))
        rows = []
        for time, time_next in zip(times[:-1], times[1:]):
            r = [float(time), float(time_next), float(self.sqrt_recip_alphas_cumprod[time]), float(self.sqrt_recipm1_alphas_cumprod[time]), 0.0, 0.0, 0.0, 0.0, 0.0]
            if time_next >= 0:
                alpha_next = self.alphas_cumprod[time_next].float().cpu()
                sigma = self.eta * ((1 - self.alphas_cumprod[time].cpu() / alpha_next) * (1 - alpha_next) / (1 - self.alphas_cumprod[time].cpu())).sqrt()
                r[8] = float(sigma)
                r[4] = float(alpha_next.sqrt())
                r[5] = float((1 - alpha_next - sigma ** 2).sqrt())
                lr = learning_rate
                if time < T * 0.05:
                    K = 0
                elif time > T * 0.9:
                    K = 3
                elif time > T * 0.75:
                    K, lr = 2, lr * 0.5
                else:
                    K, lr = 1, lr * 0.25
                r[6], r[7] = float(K), float(lr)
            rows.append(r)
        return torch.tensor(rows, dtype=torch.float32)
