"""CPU oracle for the TMDM sampler.

TEST INFRASTRUCTURE ONLY (same rule as ``nsdiff_oracle``): only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED for the sampler (epsilon network, reverse steps, loop, tile/un-tile):
``tests/golden/make_golden_tmdm.py`` runs the reference's own ``tmdm_model`` /
``tmdm_diffusion_utils`` / ``TMDM`` code on seeded inputs.  UNPINNED for the prior
``y_0_hat`` (ns-Transformer VAE mean, third-party ``torch-timeseries==0.1.10`` arithmetic that
is absent here): the prior is an INPUT of every function below.

Paths are relative to ``models/Diffusion_model/TMDM/`` in the reference checkout.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from .nsdiff_oracle import NoiseFn

EPS_NET = "model.diffussion_model."       # TMDM_model.model (TMDM) .diffussion_model (sic)


def beta_schedule(kind: str, steps: int, start: float, end: float) -> torch.Tensor:
    """make_beta_schedule (tmdm_diffusion_utils.py:6-31)."""
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
        s = 0.008
        f = lambda i: math.cos((i / steps + s) / (1 + s) * math.pi / 2) ** 2
        b = torch.tensor([min(1 - f(i + 1) / f(i), 0.999) for i in range(steps)])
        return b.flip(0) if kind == "cosine_reverse" else b
    if kind == "cosine_anneal":
        return torch.tensor([start + 0.5 * (end - start) * (1 - math.cos(t / (steps - 1) * math.pi))
                             for t in range(steps)])
    raise ValueError(kind)


def schedule_tables(steps: int, kind: str = "linear", start: float = 1e-4, end: float = 0.02) -> Dict[str, torch.Tensor]:
    """The two tables the sampler reads (TMDM.py:52-64)."""
    betas = beta_schedule(kind, steps, start, end).float()
    alphas = 1.0 - betas
    abar = alphas.cumprod(dim=0)
    s = torch.sqrt(1 - abar)
    if kind == "cosine":
        s = s * 0.9999
    return {"alphas": alphas, "one_minus_alphas_bar_sqrt": s}


def random_state_dict(nf: int, steps: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded weights in the layout of TMDM_model's sampler part (tmdm_model.py:6-36): nn.Linear default
    init, embedding rows uniform [0,1), T+1 embedding rows."""
    g = torch.Generator().manual_seed(seed)

    def linear(n_out, n_in):
        k = 1.0 / math.sqrt(n_in)
        return (torch.rand(n_out, n_in, generator=g) * 2 - 1) * k, (torch.rand(n_out, generator=g) * 2 - 1) * k

    sd = {}
    for i, n_in in ((1, 2 * nf), (2, 128), (3, 128)):
        w, b = linear(128, n_in)
        sd[EPS_NET + "lin%d.lin.weight" % i] = w
        sd[EPS_NET + "lin%d.lin.bias" % i] = b
        sd[EPS_NET + "lin%d.embed.weight" % i] = torch.rand(steps + 1, 128, generator=g)
    sd[EPS_NET + "lin4.weight"], sd[EPS_NET + "lin4.bias"] = linear(nf, 128)
    return sd


def eps_network(sd, y_t: torch.Tensor, y0hat: torch.Tensor, t: int) -> torch.Tensor:
    """ConditionalGuidedModel.forward with cat_x & cat_y_pred (tmdm_model.py:38-64): softplus, no normalise."""
    h = torch.cat((y_t, y0hat), dim=-1)
    for i in (1, 2, 3):
        p = EPS_NET + "lin%d." % i
        h = F.linear(h, sd[p + "lin.weight"], sd[p + "lin.bias"])
        h = F.softplus(sd[p + "embed.weight"][t].view(1, 1, -1) * h)
    return F.linear(h, sd[EPS_NET + "lin4.weight"], sd[EPS_NET + "lin4.bias"])


def _y0_reparam(tb, t, y, mean_T, eps):
    s = tb["one_minus_alphas_bar_sqrt"][t]
    r = (1 - s.square()).sqrt()
    return 1 / r * (y - (1 - r) * mean_T - eps * s)                     # :83-84, :99-100


def step_scalars(tb, t: int):
    """gamma_0..2 and sqrt(beta_hat) of p_sample (tmdm_diffusion_utils.py:71-88)."""
    a = tb["alphas"][t]
    s, s1 = tb["one_minus_alphas_bar_sqrt"][t], tb["one_minus_alphas_bar_sqrt"][t - 1]
    r, r1 = (1 - s.square()).sqrt(), (1 - s1.square()).sqrt()
    g0 = (1 - a) * r1 / s.square()
    g1 = s1.square() * a.sqrt() / s.square()
    g2 = 1 + (r - 1) * (a.sqrt() + r1) / s.square()
    beta_hat = s1.square() / s.square() * (1 - a)
    return g0, g1, g2, beta_hat.sqrt()


def reverse_step(sd, tb, y, y0hat, mean_T, t: int, z: torch.Tensor) -> torch.Tensor:
    """p_sample for t >= 1 with z supplied (:57-90)."""
    g0, g1, g2, sb = step_scalars(tb, t)
    y0r = _y0_reparam(tb, t, y, mean_T, eps_network(sd, y, y0hat, t))
    return g0 * y0r + g1 * y + g2 * mean_T + sb * z


def final_step(sd, tb, y, y0hat, mean_T) -> torch.Tensor:
    """p_sample_t_1to0 (:94-104)."""
    return _y0_reparam(tb, 0, y, mean_T, eps_network(sd, y, y0hat, 0))


def sample_loop(sd, tb, y0hat, mean_T, steps: int, noise: NoiseFn,
                record: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """p_sample_loop (:107-119): y_T = z + mean; per step one z drawn BEFORE the network call."""
    z = noise(tuple(mean_T.shape))
    if record is not None:
        record.append(z)
    y = z + mean_T
    for t in reversed(range(1, steps)):
        z = noise(tuple(y.shape))
        if record is not None:
            record.append(z)
        y = reverse_step(sd, tb, y, y0hat, mean_T, t, z)
    return final_step(sd, tb, y, y0hat, mean_T)


def evaluation_step(sd, tb, y0hat: torch.Tensor, steps: int, pred_len: int, n_z_samples: int,
                    parallel_sample: int, noise: NoiseFn) -> torch.Tensor:
    """TMDM_model.evaluation_step after the prior (tmdm_adapter.py:122-155).  y0hat [B, L, F] with
    L = label_len + pred_len -> [B, pred_len, F, n_z_samples]."""
    par = min(parallel_sample, n_z_samples)
    if n_z_samples % par != 0:
        raise ValueError("n_z_samples must be divisible by parallel_sample")            # :126-127
    B, L, nf = y0hat.shape
    preds = []
    for _ in range(n_z_samples // par):
        tile = y0hat.repeat(par, 1, 1, 1).transpose(0, 1).flatten(0, 1)                # node-major
        gen = sample_loop(sd, tb, tile, tile, steps, noise).reshape(B, par, L, nf)
        preds.append(gen[:, :, -pred_len:, :])
    return torch.cat(preds, dim=1).permute(0, 2, 3, 1)
