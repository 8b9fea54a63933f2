"""Noise schedule + epsilon-network container of NsDiff (host side).

Mirrors the reference's `NsDiff_net` (models/Diffusion_model/NsDiff/NsDiff_net.py:76-172):
same attribute names for the schedule tables, same parameter names under
`diffussion_model.*` (sic) so reference checkpoints load with strict=True.  The network's
arithmetic is not here: `forward` is served by the fused HIP chain kernel, which consumes
the weights and the per-step coefficient rows built by `posterior_coef_rows`.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn as nn

HIDDEN = 128
COEF_COLS = 20


def make_beta_schedule(schedule="linear", num_timesteps=1000, start=1e-5, end=1e-2) -> torch.Tensor:
    """The schedule kinds of reference nsdiff_utils.py:6-31."""
    n = num_timesteps
    if schedule == "linear":
        return torch.linspace(start, end, n)
    if schedule == "const":
        return end * torch.ones(n)
    if schedule == "quad":
        return torch.linspace(start ** 0.5, end ** 0.5, n) ** 2
    if schedule == "jsd":
        return 1.0 / torch.linspace(n, 1, n)
    if schedule == "sigmoid":
        return torch.sigmoid(torch.linspace(-6, 6, n)) * (end - start) + start
    if schedule in ("cosine", "cosine_reverse"):
        s, cap = 0.008, 0.999
        f = [math.cos((i / n + s) / (1 + s) * math.pi / 2) ** 2 for i in range(n + 1)]
        betas = torch.tensor([min(1 - f[i + 1] / f[i], cap) for i in range(n)])
        return betas.flip(0) if schedule == "cosine_reverse" else betas
    if schedule == "cosine_anneal":
        return torch.tensor([start + 0.5 * (end - start) * (1 - math.cos(t / (n - 1) * math.pi)) for t in range(n)])
    raise ValueError("unknown diffusion_schedule {!r}".format(schedule))


def nsdiff_tables(steps: int, schedule: str, beta_start: float, beta_end: float) -> Dict[str, torch.Tensor]:
    """fp32 CPU tables with the reference's names (NsDiff_net.py:92-134).

    alphas_cumprod_sum[t] (= alpha-tilde) = sum_k prod_{j=t-k..t} alpha_j and
    alphas_hat[t] = sum_k alpha_{t-k} prod_{j=t-k..t} alpha_j, evaluated with the same
    torch fp32 ops as the reference (flip / cumprod / sum) so the tables are bit-identical.
    """
    betas = make_beta_schedule(schedule, steps, beta_start, beta_end).float()
    alphas = 1.0 - betas
    cumprod = alphas.cumprod(dim=0)
    tilde, hat = torch.zeros_like(alphas), torch.zeros_like(alphas)
    for t in range(steps):
        rev = alphas[: t + 1].flip(dims=[0])
        prod = torch.cumprod(rev, dim=0)
        tilde[t] = prod.sum()
        hat[t] = (prod * rev).sum()
    betas_tilde, betas_bar = tilde - hat, 1 - cumprod
    lead = torch.ones(1)
    omabs = torch.sqrt(1 - cumprod)
    if schedule == "cosine":
        omabs = omabs * 0.9999
    return {
        "betas": betas, "alphas": alphas, "alphas_cumprod": cumprod, "alphas_cumprod_sum": tilde,
        "alphas_tilde": tilde, "alphas_hat": hat, "betas_tilde": betas_tilde, "betas_bar": betas_bar,
        "betas_tilde_m_1": torch.cat([lead, betas_tilde[:-1]]), "betas_bar_m_1": torch.cat([lead, betas_bar[:-1]]),
        "one_minus_alphas_bar_sqrt": omabs, "alphas_cumprod_prev": torch.cat([lead, cumprod[:-1]]),
        "alphas_cumprod_sum_prev": torch.cat([lead, tilde[:-1]]),
    }


def posterior_coef_rows(tb: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[T, 20] fp32 rows consumed by the chain kernel (column meaning: include/mpv_hip.h).

    Each entry is the scalar sub-expression the reference evaluates on [1,1,1] fp32 tensors
    inside p_sample / calc_gammas (nsdiff_utils.py:131-157, 80-92), in its operation order,
    so the kernel multiplies per-element values by bit-identical scalars.
    """
    a = tb["alphas"]
    bt1, bb1 = tb["betas_tilde_m_1"], tb["betas_bar_m_1"]
    bt, bb = tb["betas_tilde"], tb["betas_bar"]
    s = tb["one_minus_alphas_bar_sqrt"]
    r = (1 - s.square()).sqrt()
    lam0 = a * (1 - a) * bt1
    sp, sa = tb["alphas_cumprod_prev"].sqrt(), a.sqrt()
    cols = [
        (1 - a) ** 2 * bt1 + a * (1 - a) * (bb1 - bt1),      # c1
        a * bt1 + a * (1 - a),                               # c2
        (1 - a) ** 2,                                        # q1
        bb1 - bt1,                                           # q2
        a * bb1 - a * bt1 + (1 - a) ** 2,                    # c4
        4 * lam0, 2 * lam0,                                  # l4, l2
        bb - bt, bt,                                         # n1, bt
        1 / r, 1 - r,                                        # invr, omr
        a * (1 - a), bt1, a,                                 # aoma, bt1, a
        sp, sa, sa * (a - 1), 1 - sp,                        # sp, sa, k1, k2
    ]
    rows = torch.zeros(a.numel(), COEF_COLS, dtype=torch.float32)
    for i, c in enumerate(cols):
        rows[:, i] = c.float()
    return rows


class ConditionalLinear(nn.Module):
    """Weight holder for the reference's ConditionalLinear (denoise.py:6-20): `lin` +
    per-step scale table `embed` initialised U(0,1)."""

    def __init__(self, num_in: int, num_out: int, n_steps: int):
        super().__init__()
        self.num_out = num_out
        self.lin = nn.Linear(num_in, num_out)
        self.embed = nn.Embedding(n_steps, num_out)
        self.embed.weight.data.uniform_()


class ConditionalGuidedModel(nn.Module):
    """Weight holder for the pointwise conditional MLP (denoise.py:23-51)."""

    def __init__(self, diff_steps: int, enc_in: int):
        super().__init__()
        self.lin1 = ConditionalLinear(enc_in * 3, HIDDEN, diff_steps)
        self.lin2 = ConditionalLinear(HIDDEN, HIDDEN, diff_steps)
        self.lin3 = ConditionalLinear(HIDDEN, HIDDEN, diff_steps)
        self.lin4 = nn.Linear(HIDDEN, enc_in)
        self.sigma_lin = nn.Linear(HIDDEN, enc_in)

    def forward(self, *args, **kwargs):
        raise RuntimeError("the epsilon network runs inside the fused HIP chain kernel; "
                           "call the owning model's evaluation_step")


class NsDiff_net(nn.Module):
    def __init__(self, configs, device):
        super().__init__()
        self.args = configs
        self.device = device
        self.num_timesteps = int(configs.diffusion_steps)
        tb = nsdiff_tables(self.num_timesteps, configs.diffusion_schedule, configs.beta_start, configs.beta_end)
        if not bool((tb["betas_tilde"] >= 0).all()) or not bool(((tb["betas_bar"] - tb["betas_tilde"]) >= 0).all()):
            raise AssertionError("schedule violates betas_bar >= betas_tilde >= 0")      # NsDiff_net.py:112-114
        self._tables = tb
        for name, val in tb.items():
            setattr(self, name, val)
        self.coef_rows = posterior_coef_rows(tb)
        self.diffussion_model = ConditionalGuidedModel(self.num_timesteps, int(configs.dataset_nf))

    def native_weights(self):
        m = self.diffussion_model
        lins = (m.lin1, m.lin2, m.lin3)
        return dict(lin_w=[l.lin.weight for l in lins], lin_b=[l.lin.bias for l in lins],
                    embed=[l.embed.weight for l in lins], eps_w=m.lin4.weight, eps_b=m.lin4.bias,
                    sig_w=m.sigma_lin.weight, sig_b=m.sigma_lin.bias, coef=self.coef_rows)

    def forward(self, *args, **kwargs):
        return self.diffussion_model(*args, **kwargs)


class NsDiff_net_spatial(nn.Module):
    """Schedule tables + graph U-Net container of NsDiff_spatial (reference NsDiff_net.py:175-260): the same tables and attribute names as
    `NsDiff_net`, `diffussion_model` (sic) = NsDiff's UGnet, and the edge index the reference stores on the module (`set_edge_index`)."""

    def __init__(self, configs, device):
        super().__init__()
        from .ugnet import UGnet
        self.args = configs
        self.device = device
        self.num_timesteps = int(configs.diffusion_steps)
        tb = nsdiff_tables(self.num_timesteps, configs.diffusion_schedule, configs.beta_start, configs.beta_end)
        if not bool((tb["betas_tilde"] >= 0).all()) or not bool(((tb["betas_bar"] - tb["betas_tilde"]) >= 0).all()):
            raise AssertionError("schedule violates betas_bar >= betas_tilde >= 0")      # NsDiff_net.py:209-211
        self._tables = tb
        for name, val in tb.items():
            setattr(self, name, val)
        self.coef_rows = posterior_coef_rows(tb)
        self.diffussion_model = UGnet(configs)
        self.edge_index = None

    def set_edge_index(self, edge_index):
        self.edge_index = edge_index

    def forward(self, *args, **kwargs):
        return self.diffussion_model(*args, **kwargs)
