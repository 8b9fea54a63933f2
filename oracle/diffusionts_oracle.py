"""CPU oracle for the DiffusionTS sampling path.

TEST INFRASTRUCTURE ONLY.  No HIP path exists for this row yet: this module is the first gate of it (a restatement
pinned against the reference, to build kernels against in a later round).  Everything is a pure function of a
state dict with the reference's parameter names, so the Langevin correction's gradient (reference langevin_fn,
DiffusionTS.py:359-407) comes from autograd over the same functional forward.

Parity status: PINNED.  ``tests/golden/make_golden_diffusionts.py`` imports the reference's own DiffusionTS package
(it imports without stand-ins) and records the transformer outputs, single model predictions, a full
``fast_sample_infill`` with every normal draw, and ``evaluation_step``; ``tests/test_diffusionts_oracle.py`` checks
the functions below against them.

Paths are relative to ``models/Diffusion_model/DiffusionTS/`` in the reference checkout.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

NET = "model.model."            # DiffusionTS_model.model (Diffusion_TS) .model (Transformer)
NoiseFn = Callable[[Tuple[int, ...]], torch.Tensor]


@dataclass
class DTSConfig:
    windows: int
    pred_len: int
    nf: int
    d_model: int = 64
    n_layer_enc: int = 3
    n_layer_dec: int = 6
    n_heads: int = 4
    timesteps: int = 100
    sampling_timesteps: int = 100
    eta: float = 0.0
    n_z_samples: int = 8
    parallel_sample: int = 4
    infill_coef: float = 1e-1
    infill_learning_rate: float = 5e-2

    @property
    def seq_len(self) -> int:
        return self.windows + self.pred_len


# --------------------------------------------------------------------------------------
# schedule (DiffusionTS.py:22-32, 76-124): float64 arithmetic, buffers stored as float32
# --------------------------------------------------------------------------------------
def schedule_tables(timesteps: int) -> Dict[str, torch.Tensor]:
    x = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / timesteps) + 0.008) / 1.008 * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    f = lambda v: v.to(torch.float32)
    return {"betas": f(betas), "alphas_cumprod": f(alphas_cumprod),
            "sqrt_alphas_cumprod": f(torch.sqrt(alphas_cumprod)),
            "sqrt_one_minus_alphas_cumprod": f(torch.sqrt(1.0 - alphas_cumprod)),
            "sqrt_recip_alphas_cumprod": f(torch.sqrt(1.0 / alphas_cumprod)),
            "sqrt_recipm1_alphas_cumprod": f(torch.sqrt(1.0 / alphas_cumprod - 1))}


# --------------------------------------------------------------------------------------
# transformer (diffusionts_transformer.py:12-438, diffusionts_model_utils.py:49-202); dropout is identity in eval
# --------------------------------------------------------------------------------------
def _sinusoidal(t: torch.Tensor, dim: int) -> torch.Tensor:                       # model_utils.py:49-61
    half = dim // 2
    freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = t[:, None] * freq[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def _ada_layer_norm(sd, p: str, x, t, dim: int):                                  # model_utils.py:187-202
    emb = F.linear(F.silu(_sinusoidal(t, dim)), sd[p + "linear.weight"], sd[p + "linear.bias"]).unsqueeze(1)
    scale, shift = torch.chunk(emb, 2, dim=2)
    return F.layer_norm(x, (dim,)) * (1 + scale) + shift


def _attention(sd, p: str, x, kv, heads: int):                                    # transformer.py:144-159, 184-201
    B, T, C = x.shape
    hs = C // heads
    split = lambda v: v.view(B, -1, heads, hs).transpose(1, 2)
    k = split(F.linear(kv, sd[p + "key.weight"], sd[p + "key.bias"]))
    q = split(F.linear(x, sd[p + "query.weight"], sd[p + "query.bias"]))
    v = split(F.linear(kv, sd[p + "value.weight"], sd[p + "value.bias"]))
    att = F.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hs)), dim=-1)
    y = (att @ v).transpose(1, 2).contiguous().view(B, T, C)
    return F.linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])


def _mlp(sd, p: str, x):                                                          # Linear, GELU, Linear (transformer.py:227-232)
    return F.linear(F.gelu(F.linear(x, sd[p + "0.weight"], sd[p + "0.bias"])), sd[p + "2.weight"], sd[p + "2.bias"])


def _layer_norm(sd, p: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + "weight"], sd[p + "bias"])


def _trend_block(sd, p: str, x, seq_len: int):                                    # transformer.py:12-34
    h = F.conv1d(x, sd[p + "trend.0.weight"], sd[p + "trend.0.bias"], padding=1)           # [B, 3, d]
    h = F.gelu(h).transpose(1, 2)                                                           # [B, d, 3]
    h = F.conv1d(h, sd[p + "trend.3.weight"], sd[p + "trend.3.bias"], padding=1)           # [B, F, 3]
    lin = torch.arange(1, seq_len + 1, 1) / (seq_len + 1)
    poly = torch.stack([lin ** float(k + 1) for k in range(3)], dim=0)                      # [3, L]
    return torch.matmul(h, poly).transpose(1, 2)                                            # [B, L, F]


def _fourier_layer(x):                                                            # transformer.py:52-97 (low_freq=1, factor=1)
    b, t, d = x.shape
    xf = torch.fft.rfft(x, dim=1)
    if t % 2 == 0:
        xf, f = xf[:, 1:-1], torch.fft.rfftfreq(t)[1:-1]
    else:
        xf, f = xf[:, 1:], torch.fft.rfftfreq(t)[1:]
    top_k = int(math.log(xf.shape[1]))
    _, idx = torch.topk(xf.abs(), top_k, dim=1, largest=True, sorted=True)                  # [b, k, d]
    xf = torch.gather(xf, 1, idx)
    fk = f.view(1, -1, 1).expand(b, -1, d).gather(1, idx).unsqueeze(2)                      # [b, k, 1, d]
    xf = torch.cat([xf, xf.conj()], dim=1)
    fk = torch.cat([fk, -fk], dim=1)
    tt = torch.arange(t, dtype=torch.float).view(1, 1, t, 1)
    amp, phase = xf.abs().unsqueeze(2), xf.angle().unsqueeze(2)
    return (amp * torch.cos(2 * math.pi * fk * tt + phase)).sum(dim=1)                      # [b, t, d]


def transformer(sd, cfg: DTSConfig, x: torch.Tensor, t: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(trend, season_error) = Transformer.forward(x [B, L, F], t [B]) (transformer.py:422-438)."""
    d, L = cfg.d_model, cfg.seq_len
    conv_mlp = lambda p, v: F.conv1d(v.transpose(1, 2), sd[p + "sequential.1.weight"], sd[p + "sequential.1.bias"], padding=1).transpose(1, 2)
    emb = conv_mlp(NET + "emb.", x)
    h = emb + sd[NET + "pos_enc.pe"]
    for i in range(cfg.n_layer_enc):                                              # EncoderBlock :235-239
        p = NET + "encoder.blocks.%d." % i
        n1 = _ada_layer_norm(sd, p + "ln1.", h, t, d)
        h = h + _attention(sd, p + "attn.", n1, n1, cfg.n_heads)
        h = h + _mlp(sd, p + "mlp.", _layer_norm(sd, p + "ln2.", h))
    enc = h
    h = emb + sd[NET + "pos_dec.pe"]
    season = torch.zeros(x.shape[0], L, d)
    trend = torch.zeros(x.shape[0], L, cfg.nf)
    means = []
    for i in range(cfg.n_layer_dec):                                              # DecoderBlock :323-332
        p = NET + "decoder.blocks.%d." % i
        n1 = _ada_layer_norm(sd, p + "ln1.", h, t, d)
        h = h + _attention(sd, p + "attn1.", n1, n1, cfg.n_heads)
        h = h + _attention(sd, p + "attn2.", _ada_layer_norm(sd, p + "ln1_1.", h, t, d), enc, cfg.n_heads)
        x1, x2 = F.conv1d(h, sd[p + "proj.weight"], sd[p + "proj.bias"]).chunk(2, dim=1)
        trend = trend + _trend_block(sd, p + "trend.", x1, L)
        season = season + _fourier_layer(x2)
        h = h + _mlp(sd, p + "mlp.", _layer_norm(sd, p + "ln2.", h))
        m = torch.mean(h, dim=1, keepdim=True)
        means.append(F.linear(m, sd[p + "linear.weight"], sd[p + "linear.bias"]))
        h = h - m
    mean = torch.cat(means, dim=1)                                                # [B, n_dec, F]
    res = conv_mlp(NET + "inverse.", h)
    res_m = torch.mean(res, dim=1, keepdim=True)
    w = sd[NET + "combine_s.weight"]
    pad = (w.shape[-1] - 1) // 2
    s_in = season.transpose(1, 2)
    if pad:
        s_in = F.pad(s_in, (pad, pad), mode="circular")
    season_error = F.conv1d(s_in, w).transpose(1, 2) + res - res_m
    trend = F.conv1d(mean, sd[NET + "combine_m.weight"]) + res_m + trend
    return trend, season_error


def model_output(sd, cfg, x, t):                                                  # Diffusion_TS.output :147-150
    trend, season = transformer(sd, cfg, x, t)
    return trend + season


def model_predictions(sd, cfg, tb, x, t, clip_x_start: bool):                      # :152-160, :126-130
    x_start = model_output(sd, cfg, x, t)
    if clip_x_start:
        x_start = torch.clamp(x_start, min=-1.0, max=1.0)
    a = tb["sqrt_recip_alphas_cumprod"][t].view(-1, 1, 1)
    b = tb["sqrt_recipm1_alphas_cumprod"][t].view(-1, 1, 1)
    return (a * x - x_start) / b, x_start


def q_sample(tb, x_start, t, noise):                                              # :232-237
    return tb["sqrt_alphas_cumprod"][t].view(-1, 1, 1) * x_start + tb["sqrt_one_minus_alphas_cumprod"][t].view(-1, 1, 1) * noise


def langevin(sd, cfg, tb, sample, mean, sigma, t, target, mask, noise: NoiseFn, record, batch_of: Optional[int] = None):    # :359-407
    """K Adagrad steps (K from the time index) on the un-masked part of `sample`; each step also draws an `epsilon`
    that is multiplied by coef_ = 0 (drawn all the same: the draw order matters for noise replay).

    `batch_of` (test aid): the rows are a SUBSET of a reference batch of that many sequences -- the reference's two batch means
    (`.mean(dim=0)` over the batch, and over all masked elements of the batch) then divide the subset's sums by the FULL batch's
    counts, which is what these rows see inside the full batch (sequences couple through nothing else)."""
    t0 = int(t[0].item())
    lr = cfg.infill_learning_rate
    if t0 < cfg.timesteps * 0.05:
        K = 0
    elif t0 > cfg.timesteps * 0.9:
        K = 3
    elif t0 > cfg.timesteps * 0.75:
        K, lr = 2, lr * 0.5
    else:
        K, lr = 1, lr * 0.25
    param = sample.clone()
    for _ in range(K):
        p = param.detach().clone().requires_grad_(True)
        with torch.enable_grad():
            x_start = model_output(sd, cfg, p, t)
            infill = (x_start[mask] - target[mask]) ** 2
            if float(sigma.mean()) == 0 and batch_of is not None:
                logp = cfg.infill_coef * (((mean - p) ** 2 / 1.0).sum(dim=0) / float(batch_of)).sum()
                infill = infill.sum() / float(batch_of * int(mask[0].sum()))
            elif float(sigma.mean()) == 0:
                logp = cfg.infill_coef * ((mean - p) ** 2 / 1.0).mean(dim=0).sum()
                infill = infill.mean(dim=0).sum()
            else:
                logp = cfg.infill_coef * ((mean - p) ** 2 / sigma).mean(dim=0).sum()
                infill = (infill / sigma.mean()).mean(dim=0).sum()
            (g,) = torch.autograd.grad(logp + infill, p)
        # a fresh Adagrad every iteration (:384): state_sum = g^2, step = lr * g / (sqrt(g^2) + 1e-10)
        new = p.detach() - lr * g / (g.pow(2).sqrt() + 1e-10)
        eps = noise(tuple(new.shape))
        if record is not None:
            record.append(eps)
        param = new + 0.0 * float(sigma.mean()) * eps
    out = sample.clone()
    out[~mask] = param[~mask]
    return out


def fast_sample_infill(sd, cfg: DTSConfig, tb, target, mask, noise: NoiseFn, record: Optional[List[torch.Tensor]] = None,
                       batch_of: Optional[int] = None):
    """DDIM sampler with infilling (DiffusionTS.py:277-310).  Draw order per pair with time_next >= 0: `noise`, then
    K Langevin `epsilon`s, then q_sample's noise."""
    def draw(shape):
        z = noise(shape)
        if record is not None:
            record.append(z)
        return z

    batch = target.shape[0]
    times = list(reversed(torch.linspace(-1, cfg.timesteps - 1, steps=cfg.sampling_timesteps + 1).int().tolist()))
    img = draw(tuple(target.shape))
    for time, time_next in zip(times[:-1], times[1:]):
        t = torch.full((batch,), time, dtype=torch.long)
        with torch.no_grad():
            pred_noise, x_start = model_predictions(sd, cfg, tb, img, t, clip_x_start=True)
        if time_next < 0:
            img = x_start
            continue
        alpha, alpha_next = tb["alphas_cumprod"][time], tb["alphas_cumprod"][time_next]
        sigma = cfg.eta * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        pred_mean = x_start * alpha_next.sqrt() + c * pred_noise
        img = pred_mean + sigma * draw(tuple(img.shape))
        img = langevin(sd, cfg, tb, img, pred_mean, sigma, t, target, mask, noise, record, batch_of=batch_of)
        target_t = q_sample(tb, target, t, draw(tuple(target.shape)))
        img = torch.where(mask, target_t, img)
    return torch.where(mask, target, img)


def evaluation_step(sd, cfg: DTSConfig, batch: torch.Tensor, noise: NoiseFn, record=None) -> torch.Tensor:
    """DiffusionTS_model.evaluation_step (DiffusionTS_model.py:72-109): history + zero future, SAMPLE-major tiling
    (`x.repeat(par, 1, 1)`) but node-major un-tiling (`reshape(B, par, ...)`) -- reproduced as is.
    -> [B, pred_len, F, n_z_samples]."""
    par = min(cfg.parallel_sample, cfg.n_z_samples)
    if cfg.n_z_samples % par != 0:
        raise ValueError("n_z_samples must be divisible by parallel_sample")
    tb = schedule_tables(cfg.timesteps)
    bx = batch[:, : cfg.windows, :]
    B = bx.shape[0]
    x = torch.cat([bx, torch.zeros(B, cfg.pred_len, cfg.nf)], dim=1)
    gt = torch.cat([torch.ones(cfg.windows, cfg.nf, dtype=torch.bool), torch.zeros(cfg.pred_len, cfg.nf, dtype=torch.bool)], dim=0)
    mask = gt.expand(B, -1, -1)
    samples = []
    for _ in range(cfg.n_z_samples // par):
        rx, rm = x.repeat(par, 1, 1), mask.repeat(par, 1, 1)
        s = fast_sample_infill(sd, cfg, tb, rx * rm, rm, noise, record)
        samples.append(s[:, -cfg.pred_len:, :].reshape(B, par, cfg.pred_len, cfg.nf))
    preds = torch.cat(samples, dim=1)
    return preds.reshape(B, cfg.n_z_samples, cfg.pred_len, cfg.nf).permute(0, 2, 3, 1)
