"""CPU oracle for the NsDiff rolling-window MPV sampling path.

TEST INFRASTRUCTURE ONLY.  This module is a plain-PyTorch (CPU, fp32) restatement of
the reference algorithm.  It exists to *check* the HIP path: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The product package must never route through it.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the reference's own
Python modules (in the build container only) and records their outputs on seeded
inputs; ``tests/test_oracle_golden.py`` checks every function below against those
vectors (``-m "not gpu"``).

Every function cites the reference file:line it follows (paths relative to the
reference checkout).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

EPS = 10e-8  # NsDiff_model.py:39 / nsdiff_utils.py:5 (sic: 1e-7)


# --------------------------------------------------------------------------------------
# a7. schedule tables  (NsDiff_net.py:92-134, nsdiff_utils.py:6-31)
# --------------------------------------------------------------------------------------
def beta_schedule(kind: str, steps: int, start: float, end: float) -> torch.Tensor:
    """nsdiff_utils.py:6-31 (the schedules the NsDiff YAMLs can name)."""
    if kind == "linear":
        return torch.linspace(start, end, steps)
    if kind == "const":
        return end * torch.ones(steps)
    if kind == "quad":
        return torch.linspace(start ** 0.5, end ** 0.5, steps) ** 2
    if kind == "jsd":
        return 1.0 / torch.linspace(steps, 1, steps)
    if kind == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, steps)) * (end - start) + start
    if kind in ("cosine", "cosine_reverse"):
        s, cap = 0.008, 0.999

        def abar(i):
            return math.cos((i / steps + s) / (1 + s) * math.pi / 2) ** 2

        b = torch.tensor([min(1 - abar(i + 1) / abar(i), cap) for i in range(steps)])
        return b.flip(0) if kind == "cosine_reverse" else b
    if kind == "cosine_anneal":
        return torch.tensor([start + 0.5 * (end - start) * (1 - math.cos(t / (steps - 1) * math.pi))
                             for t in range(steps)])
    raise ValueError("unknown diffusion_schedule {!r}".format(kind))


def schedule_tables(steps: int, kind: str = "linear", start: float = 1e-4, end: float = 0.02) -> Dict[str, torch.Tensor]:
    """All per-timestep fp32 tables NsDiff_net.__init__ builds (NsDiff_net.py:92-134).

    alphas_tilde[t] = sum_k prod_{j=t-k..t} alpha_j           (compute_tilde_alpha :34-43)
    alphas_hat[t]   = sum_k alpha_{t-k} prod_{j=t-k..t} alpha_j (compute_hat_alpha :45-54)
    """
    betas = beta_schedule(kind, steps, start, end).float()
    alphas = 1.0 - betas
    abar = alphas.cumprod(dim=0)
    tilde = torch.zeros_like(alphas)
    hat = torch.zeros_like(alphas)
    for t in range(steps):
        back = alphas[: t + 1].flip(dims=[0])          # a_t, a_{t-1}, ...
        run = torch.cumprod(back, dim=0)
        tilde[t] = run.sum()
        hat[t] = (run * back).sum()
    betas_tilde = tilde - hat
    betas_bar = 1 - abar
    one = torch.ones(1)
    omabs = torch.sqrt(1 - abar)
    if kind == "cosine":
        omabs = omabs * 0.9999                           # NsDiff_net.py:124-125
    return {
        "betas": betas,
        "alphas": alphas,
        "alphas_cumprod": abar,
        "alphas_cumprod_sum": tilde,
        "alphas_hat": hat,
        "betas_tilde": betas_tilde,
        "betas_bar": betas_bar,
        "betas_tilde_m_1": torch.cat([one, betas_tilde[:-1]]),
        "betas_bar_m_1": torch.cat([one, betas_bar[:-1]]),
        "one_minus_alphas_bar_sqrt": omabs,
        "alphas_cumprod_prev": torch.cat([one, abar[:-1]]),
        "alphas_cumprod_sum_prev": torch.cat([one, tilde[:-1]]),
    }


# --------------------------------------------------------------------------------------
# parameter container: tensors keyed exactly like the reference state_dict
# --------------------------------------------------------------------------------------
EPS_NET = "model.diffussion_model."   # (sic) NsDiff_net.py:157
G_NET = "cond_pred_model_g.mlp."      # g_backbone.py:32-40


def random_state_dict(nf: int, steps: int, windows: int, pred_len: int, rolling: int,
                      seed: int = 123, with_g: bool = True) -> Dict[str, torch.Tensor]:
    """Seeded default-init weights with the reference's shapes and init distributions.

    nn.Linear default init = kaiming_uniform(a=sqrt(5)) -> U(-1/sqrt(in), 1/sqrt(in)) for
    weight and bias; ConditionalLinear.embed.weight ~ U(0,1) (denoise.py:11-12);
    LayerNorm weight=1, bias=0.  The draw order here is the oracle's own (it does not try
    to match torch's module-construction RNG stream; goldens carry their weights).
    """
    g = torch.Generator().manual_seed(seed)

    def lin(n_out, n_in):
        bound = 1.0 / math.sqrt(n_in)
        w = (torch.rand(n_out, n_in, generator=g) * 2 - 1) * bound
        b = (torch.rand(n_out, generator=g) * 2 - 1) * bound
        return w, b

    sd: Dict[str, torch.Tensor] = {"scaler_mean": torch.zeros(nf), "scaler_std": torch.ones(nf)}
    dims = [(128, 3 * nf), (128, 128), (128, 128)]
    for i, (o, k) in enumerate(dims, start=1):
        w, b = lin(o, k)
        sd[EPS_NET + "lin%d.lin.weight" % i] = w
        sd[EPS_NET + "lin%d.lin.bias" % i] = b
        sd[EPS_NET + "lin%d.embed.weight" % i] = torch.rand(steps, o, generator=g)
    for name in ("lin4", "sigma_lin"):
        w, b = lin(nf, 128)
        sd[EPS_NET + name + ".weight"] = w
        sd[EPS_NET + name + ".bias"] = b
    if with_g:
        hidden = 512
        for idx, (o, k) in zip((0, 3, 6), ((hidden, windows - rolling), (hidden, hidden), (pred_len, hidden))):
            w, b = lin(o, k)
            sd[G_NET + "%d.weight" % idx] = w
            sd[G_NET + "%d.bias" % idx] = b
        for idx in (2, 5):
            sd[G_NET + "%d.weight" % idx] = torch.ones(nf, hidden)
            sd[G_NET + "%d.bias" % idx] = torch.zeros(nf, hidden)
    return sd


# --------------------------------------------------------------------------------------
# a8. epsilon network  (denoise.py:6-51, NsDiff_net.py:163-172)
# --------------------------------------------------------------------------------------
def eps_network(sd: Dict[str, torch.Tensor], y_t: torch.Tensor, y0hat: torch.Tensor,
                gx: torch.Tensor, t: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(eps, sigma_theta) = ConditionalGuidedModel(y_t, y0hat, gx, t); all [N, O, F].

    h = cat(y_t, y0hat, gx); three times: h = normalize(softplus(embed_i[t] * (W_i h + b_i)));
    eps = W4 h + b4; sigma = softplus(W_s softplus(h) + b_s).  The same t for every row
    (NsDiff_net.py:166-167 repeats the scalar step index over the batch).
    """
    h = torch.cat((y_t, y0hat, gx), dim=-1)
    for i in (1, 2, 3):
        p = EPS_NET + "lin%d." % i
        h = F.linear(h, sd[p + "lin.weight"], sd[p + "lin.bias"])
        h = sd[p + "embed.weight"][t].view(1, 1, -1) * h
        h = F.normalize(F.softplus(h), dim=-1)
    eps = F.linear(h, sd[EPS_NET + "lin4.weight"], sd[EPS_NET + "lin4.bias"])
    sig = F.softplus(F.linear(F.softplus(h), sd[EPS_NET + "sigma_lin.weight"], sd[EPS_NET + "sigma_lin.bias"]))
    return eps, sig


# --------------------------------------------------------------------------------------
# a9-a11. reverse process  (nsdiff_utils.py:34-56, 80-92, 111-158, 209-239, 271-284)
# --------------------------------------------------------------------------------------
def _sigma_y0_and_noise(tb, t, gx, sigma_theta):
    """Quadratic for Sigma_hat_{Y0} and the forward-noise variance (nsdiff_utils.py:138-147).

    A negative discriminant yields NaN exactly as the reference does (no clamp).
    """
    a = tb["alphas"][t]
    bt1, bb1 = tb["betas_tilde_m_1"][t], tb["betas_bar_m_1"][t]
    bt, bb = tb["betas_tilde"][t], tb["betas_bar"][t]
    lam0 = a * (1 - a) * bt1
    lam1 = ((1 - a) ** 2 * bt1 + a * (1 - a) * (bb1 - bt1)) * gx - sigma_theta * (a * bt1 + a * (1 - a))
    lam2 = gx ** 2 * (1 - a) ** 2 * (bb1 - bt1) - sigma_theta * gx * (a * bb1 - a * bt1 + (1 - a) ** 2)
    sigma_y0 = (-lam1 + (lam1 ** 2 - 4 * lam0 * lam2).sqrt()) / (2 * lam0)
    noise = (bb - bt) * gx + bt * sigma_y0
    return sigma_y0, noise


def _y0_reparam(tb, t, y, mean_T, eps, noise):
    s = tb["one_minus_alphas_bar_sqrt"][t]
    r = (1 - s.square()).sqrt()                           # nsdiff_utils.py:135
    return 1 / r * (y - (1 - r) * mean_T - eps * torch.sqrt(noise))


def reverse_step(sd, tb, y, y0hat, gx, mean_T, t: int, z: torch.Tensor, net=None) -> torch.Tensor:
    """One p_sample step t>=1 with the noise z supplied (nsdiff_utils.py:111-158).  `net(sd, y, y0hat, gx, t) -> (eps, sigma)`: the
    epsilon network, the pointwise MLP by default (NsDiff_spatial passes its graph U-Net, oracle/nsdiff_spatial_oracle.py)."""
    eps, sig = (net or eps_network)(sd, y, y0hat, gx, t)
    sigma_y0, noise = _sigma_y0_and_noise(tb, t, gx, sig)
    y0r = _y0_reparam(tb, t, y, mean_T, eps, noise)
    # calc_gammas / cal_sigma12 (nsdiff_utils.py:40-56, 80-92)
    a = tb["alphas"][t]
    bt1, bb1 = tb["betas_tilde_m_1"][t], tb["betas_bar_m_1"][t]
    s1 = (1 - a) ** 2 * gx + a * (1 - a) * sigma_y0
    s2 = (bb1 - bt1) * gx + bt1 * sigma_y0
    root_a = a.sqrt()
    root_prev = tb["alphas_cumprod_prev"][t].sqrt()
    den = a * s2 + s1
    g0 = root_prev * s1 / den
    g1 = root_a * s2 / den
    g2 = ((root_a * (a - 1)) * s2 + (1 - root_prev) * s1) / den
    return g0 * y0r + g1 * y + g2 * mean_T + torch.sqrt(sig) * z


def final_step(sd, tb, y, y0hat, gx, mean_T, net=None) -> torch.Tensor:
    """p_sample_t_1to0: returns the y0 re-parameterisation at t=0, no noise (:209-239)."""
    eps, sig = (net or eps_network)(sd, y, y0hat, gx, 0)
    _, noise = _sigma_y0_and_noise(tb, 0, gx, sig)
    return _y0_reparam(tb, 0, y, mean_T, eps, noise)


NoiseFn = Callable[[Tuple[int, ...]], torch.Tensor]


def torch_cpu_noise() -> NoiseFn:
    """Noise drawn from torch's default CPU generator, exactly what ``torch.randn_like``
    consumes in the reference when it runs on CPU (nsdiff_utils.py:130,273)."""
    return lambda shape: torch.randn(shape)


def replay_noise(draws: Sequence[torch.Tensor]) -> NoiseFn:
    it = iter(draws)
    return lambda shape: next(it).reshape(shape)


def sample_loop(sd, tb, y0hat, gx, mean_T, steps: int, noise: NoiseFn,
                record: Optional[List[torch.Tensor]] = None, net=None) -> torch.Tensor:
    """p_sample_loop (:271-284).  Draw order: z_T, then one z per step t=T-1..1 taken
    AFTER that step's network call; the final step draws nothing."""
    z = noise(tuple(mean_T.shape))
    if record is not None:
        record.append(z)
    y = torch.sqrt(gx) * z + mean_T
    for t in reversed(range(1, steps)):
        z = noise(tuple(y.shape))
        if record is not None:
            record.append(z)
        y = reverse_step(sd, tb, y, y0hat, gx, mean_T, t, z, net=net)
    return final_step(sd, tb, y, y0hat, gx, mean_T, net=net)


# --------------------------------------------------------------------------------------
# a4, a5. sigma head g(x)  (sigma.py:34-72, g_backbone.py:49-72)
# --------------------------------------------------------------------------------------
def trailing_variance(x: torch.Tensor, k: int) -> torch.Tensor:
    """Population variance over the trailing k samples, first sample replicated k-1 times
    at the front (sigma.py:34-72).  x: [B, T, F] -> [B, T, F]."""
    if x.dim() != 3:
        raise ValueError("x_enc must be a 3D tensor with shape (B, T, N)")
    if k < 1 or k > x.shape[1]:
        raise ValueError("window_size must be between 1 and T")
    padded = F.pad(x, (0, 0, k - 1, 0), mode="replicate")
    return padded.unfold(1, k, 1).var(dim=3, unbiased=False)


def sigma_head(sd, x: torch.Tensor, k: int, pred_len: int) -> torch.Tensor:
    """SigmaEstimation.forward (g_backbone.py:49-72): [B, T, F] -> [B, pred_len, F]."""
    T = x.shape[1]
    v = trailing_variance(x, k)[:, -(T - k):, :] + 10e-8
    h = v.permute(0, 2, 1)                                # [B, F, T-k]
    h = F.relu(F.linear(h, sd[G_NET + "0.weight"], sd[G_NET + "0.bias"]))
    h = F.layer_norm(h, tuple(sd[G_NET + "2.weight"].shape), sd[G_NET + "2.weight"], sd[G_NET + "2.bias"])
    h = F.relu(F.linear(h, sd[G_NET + "3.weight"], sd[G_NET + "3.bias"]))
    h = F.layer_norm(h, tuple(sd[G_NET + "5.weight"].shape), sd[G_NET + "5.weight"], sd[G_NET + "5.bias"])
    h = F.linear(h, sd[G_NET + "6.weight"], sd[G_NET + "6.bias"])
    return F.softplus(h).permute(0, 2, 1)[:, -pred_len:, :]


# --------------------------------------------------------------------------------------
# a12. evaluation_step  (NsDiff_model.py:180-268 and :404-495)
# --------------------------------------------------------------------------------------
@dataclass
class NsDiffConfig:
    windows: int
    pred_len: int
    nf: int
    rolling: int
    steps: int
    n_z_samples: int
    parallel_sample: int
    variant: str = "cond_var"      # cond_var | Guassian (sic) | with y0hat supplied: cond_mean | wo_UANS | NsDiff_model
    schedule: str = "linear"
    beta_start: float = 1e-4
    beta_end: float = 0.02
    tables: Dict[str, torch.Tensor] = field(default_factory=dict)

    def __post_init__(self):
        if not self.tables:
            self.tables = schedule_tables(self.steps, self.schedule, self.beta_start, self.beta_end)


def _tile_node_major(x: torch.Tensor, reps: int) -> torch.Tensor:
    """[B,O,F] -> [B*reps,O,F], node-major / sample-minor (NsDiff_model.py:229-236)."""
    B, O, nf = x.shape
    return x.unsqueeze(1).expand(B, reps, O, nf).reshape(B * reps, O, nf)


def evaluation_step(sd, cfg: NsDiffConfig, batch: torch.Tensor, noise: Optional[NoiseFn] = None,
                    y0hat: Optional[torch.Tensor] = None,
                    record: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """Scaled window [B, >=windows, F] -> samples [B, pred_len, F, n_z_samples]."""
    noise = noise or torch_cpu_noise()
    x = batch[:, : cfg.windows, :]
    B = x.shape[0]
    if cfg.variant in ("cond_var", "Guassian"):
        y0hat = torch.zeros(B, cfg.pred_len, x.shape[2])                 # :442-446
    elif y0hat is None:
        raise ValueError("variant {!r} needs y0hat = f(x) supplied".format(cfg.variant))
    if cfg.variant in ("Guassian", "cond_mean"):
        gx = torch.ones(B, cfg.pred_len, x.shape[2])                      # :457-458 (no sigma head: :296-303)
    elif cfg.variant == "NsDiff_model":
        gx = sigma_head(sd, x, cfg.rolling, cfg.pred_len)                 # :223 (no +EPS)
    else:
        gx = sigma_head(sd, x, cfg.rolling, cfg.pred_len) + EPS           # :450
    P = int(cfg.parallel_sample)
    chunks = []
    for _ in range(cfg.n_z_samples // P):                                 # :227 (floors)
        y0_t = _tile_node_major(y0hat, P)
        g_t = _tile_node_major(gx, P)
        y = sample_loop(sd, cfg.tables, y0_t, g_t, y0_t, cfg.steps, noise, record)
        chunks.append(y.reshape(B, P, cfg.pred_len, cfg.nf))
    preds = torch.cat(chunks, dim=1)
    outs = preds.permute(0, 2, 3, 1)
    assert tuple(outs.shape[1:]) == (cfg.pred_len, cfg.nf, cfg.n_z_samples)
    return outs


# --------------------------------------------------------------------------------------
# a1, a2. sub-sampling and window arithmetic  (diffusion_model_uncertainy.py:119-160, 467-483)
# --------------------------------------------------------------------------------------
def sampling_stride(sampling_t) -> int:
    if sampling_t is None or sampling_t <= 0.1:
        return 1
    return max(1, int(sampling_t / 0.1))


def window_count(length: int, windows: int, step: int) -> int:
    return 0 if length < windows else (length - windows) // step + 1


def window_starts(length: int, windows: int, step: int) -> np.ndarray:
    return np.arange(window_count(length, windows, step), dtype=np.int64) * step


def sliding_windows(series: torch.Tensor, time: np.ndarray, windows: int, step: int):
    """[V, L, F] -> list of W tensors [V, windows, F] and their time stamps (:137-147)."""
    if series.ndim != 3:
        raise ValueError("torch_time_series must have shape [Node_num, T_obs_num, F].")
    if series.shape[1] < windows:
        raise ValueError("T_obs_num ({}) is shorter than windows ({}).".format(series.shape[1], windows))
    starts = window_starts(series.shape[1], windows, step)
    wins = [series[:, s: s + windows, :] for s in starts]
    return wins, np.asarray(time)[windows - 1:: step]


# --------------------------------------------------------------------------------------
# a3, a14. scaler and MPV summaries  (diffusion_model_uncertainy.py:267-320, 529-550)
# --------------------------------------------------------------------------------------
def scale(x, mean, std):
    return (x - mean) / std


def mpv_network(pred: torch.Tensor, mean: Optional[torch.Tensor] = None,
                std: Optional[torch.Tensor] = None) -> Tuple[float, float]:
    """[V,O,F,S] -> (pred_mean, MPV): inverse-scale on dim -2, population variance over
    samples, mean over everything (:286-303)."""
    if mean is not None:
        shape = [1] * pred.ndim
        shape[-2] = mean.numel()
        pred = pred * std.view(*shape) + mean.view(*shape)
    return float(pred.mean()), float(pred.var(dim=-1, unbiased=False).mean())


def mpv_slbp(pred: torch.Tensor, pred_dim: int = 0) -> float:
    """[O,F,S] -> var over samples, mean over horizon, one feature; no inverse scaling
    (:529-538)."""
    return float(pred.var(dim=-1, unbiased=False).mean(dim=0)[pred_dim])


def gx_summary(gx: torch.Tensor, pred_dim: int = 0) -> Tuple[float, float]:
    """[V,O,F] -> (mean of all, mean over nodes of the horizon-mean of one feature) (:306-320)."""
    return float(gx.mean()), float(gx.mean(dim=1)[:, pred_dim].mean())


# --------------------------------------------------------------------------------------
# whole-curve driver used by bench.py's cpu_baseline leg and by the parity tests
# --------------------------------------------------------------------------------------
def mpv_curve(sd, cfg: NsDiffConfig, series: torch.Tensor, windows_step: int,
              noise: Optional[NoiseFn] = None, max_windows: Optional[int] = None,
              use_scaler: bool = True):
    """Sequential window loop of run_evaluation_cache (:323-339) + summarize (:286-303).

    series: already sub-sampled [V, L, F].  Returns (list of samples, pred_mean[], mpv[]).
    """
    wins, _ = sliding_windows(series, np.arange(series.shape[1]), cfg.windows, windows_step)
    if max_windows is not None:
        wins = wins[:max_windows]
    mean, std = sd["scaler_mean"], sd["scaler_std"]
    preds, means, mpvs = [], [], []
    for w in wins:
        x = scale(w, mean, std) if use_scaler else w
        out = evaluation_step(sd, cfg, x.clone(), noise)[:, -cfg.pred_len:, :, :]
        preds.append(out)
        m, v = mpv_network(out, mean if use_scaler else None, std if use_scaler else None)
        means.append(m)
        mpvs.append(v)
    return preds, means, mpvs
