"""Host-side tables of the TMDM sampler.

Mirrors reference models/Diffusion_model/TMDM/tmdm_diffusion_utils.py: `make_beta_schedule` (:6-31) and the
per-step scalars `p_sample` / `p_sample_t_1to0` derive from `alphas` and `one_minus_alphas_bar_sqrt`
(:69-88, :96-100).  The reverse loop itself (`p_sample_loop`, :107-119) runs inside the fused HIP chain
kernel; there is no torch implementation of it here.
"""
from __future__ import annotations

import torch

from ..NsDiff.NsDiff_net import COEF_COLS, make_beta_schedule  # same schedule kinds as nsdiff_utils.py:6-31

__all__ = ["make_beta_schedule", "tmdm_tables", "tmdm_coef_rows", "p_sample_loop"]


def tmdm_tables(steps: int, schedule: str, beta_start: float, beta_end: float):
    """fp32 CPU tables with the reference's names (TMDM.py:52-72)."""
    betas = make_beta_schedule(schedule, steps, beta_start, beta_end).float()
    alphas = 1.0 - betas
    cumprod = alphas.cumprod(dim=0)
    omabs = torch.sqrt(1 - cumprod)
    if schedule == "cosine":
        omabs = omabs * 0.9999
    prev = torch.cat([torch.ones(1), cumprod[:-1]])
    return {
        "betas": betas, "betas_sqrt": torch.sqrt(betas), "alphas": alphas, "one_minus_betas_sqrt": torch.sqrt(alphas),
        "alphas_bar_sqrt": torch.sqrt(cumprod), "one_minus_alphas_bar_sqrt": omabs, "alphas_cumprod_prev": prev,
        "posterior_mean_coeff_1": betas * torch.sqrt(prev) / (1.0 - cumprod),
        "posterior_mean_coeff_2": torch.sqrt(alphas) * (1 - prev) / (1 - cumprod),
        "posterior_variance": betas * (1.0 - prev) / (1.0 - cumprod),
    }


def tmdm_coef_rows(tb) -> torch.Tensor:
    """[T, 20] fp32 rows for the chain kernel (column meaning: include/mpv_hip.h, TMDM section).  Each entry
    is the scalar the reference evaluates on [1,1,1] fp32 tensors, in its operation order.  Row 0 holds only
    the y0 re-parameterisation scalars (the final step has no gammas)."""
    a = tb["alphas"]
    s = tb["one_minus_alphas_bar_sqrt"]
    s1 = torch.cat([s[:1], s[:-1]])                      # index t-1 (row 0 unused)
    r, r1 = (1 - s.square()).sqrt(), (1 - s1.square()).sqrt()
    g0 = (1 - a) * r1 / s.square()
    g1 = s1.square() * a.sqrt() / s.square()
    g2 = 1 + (r - 1) * (a.sqrt() + r1) / s.square()
    beta_hat = s1.square() / s.square() * (1 - a)
    rows = torch.zeros(a.numel(), COEF_COLS, dtype=torch.float32)
    for i, c in enumerate((1 / r, 1 - r, s, g0, g1, g2, beta_hat.sqrt())):
        rows[:, i] = c.float()
    rows[0, 3:7] = 0
    return rows


def p_sample_loop(*args, **kwargs):
    raise RuntimeError("the TMDM reverse loop runs inside the fused HIP chain kernel; "
                       "call TMDM_model.evaluation_step / sample_windows")
