"""Noise schedule tables and the inference time schedule of DiffSTG (mirrors reference DiffSTG/diffusion_schedulers.py:39-64,
99-125): float64 NumPy, exactly the reference's expressions -- the sampler applies them as Python floats (`.item()`)."""
from __future__ import annotations

import math

import numpy as np


class GaussianDiffusion(object):
    def __init__(self, T, schedule, loss_weight_schedule="constant"):
        self.T = T
        self.loss_weight_schedule = loss_weight_schedule
        if schedule == "linear":
            self.beta = np.linspace(1e-4, 2e-2, T)
        elif schedule == "quad":
            self.beta = np.linspace(1e-4 ** 0.5, 2e-2 ** 5, T) ** 2                 # (sic) reference :54
        elif schedule == "cosine":
            self.alphabar = self._cos_noise(np.arange(0, T + 1, 1)) / self._cos_noise(0)
            self.beta = np.clip(1 - (self.alphabar[1:] / self.alphabar[:-1]), None, 0.999)
        else:
            raise AttributeError("unknown diffusion schedule {!r}".format(schedule))
        self.betabar = np.cumprod(self.beta)
        self.alpha = np.concatenate((np.array([1.0]), 1 - self.beta))
        self.alphabar = np.cumprod(self.alpha)

    def _cos_noise(self, t):
        offset = 0.008
        return np.cos(math.pi * 0.5 * (t / self.T + offset) / (1 + offset)) ** 2


class InferenceSchedule(object):
    def __init__(self, inference_schedule="linear", T=1000, inference_T=1000):
        self.inference_schedule, self.T, self.inference_T = inference_schedule, T, inference_T

    def __call__(self, i):
        assert 0 <= i < self.inference_T
        if self.inference_schedule == "linear":
            t1 = self.T - int((float(i) / self.inference_T) * self.T)
            t2 = self.T - int((float(i + 1) / self.inference_T) * self.T)
        elif self.inference_schedule == "cosine":
            t1 = self.T - int(np.sin((float(i) / self.inference_T) * np.pi / 2) * self.T)
            t2 = self.T - int(np.sin((float(i + 1) / self.inference_T) * np.pi / 2) * self.T)
        else:
            raise ValueError("Unknown inference schedule: {}".format(self.inference_schedule))
        return np.clip(t1, 1, self.T), np.clip(t2, 0, self.T - 1)


def posterior_plan(diffusion: GaussianDiffusion, inference_schedule: str, steps: int, trick: str) -> np.ndarray:
    """One row per inference step, float64 [steps][6]: 0 t1, 1 ddpm flag, then the three Python-float coefficients of
    gaussian_posterior (reference graph_diffusion_model.py:46-73):
      ddim: x' = c1 (x - c2 eps) + c3 eps,  c1 = sqrt(abar_t2 / abar_t1), c2 = sqrt(1 - abar_t1), c3 = sqrt(1 - abar_t2)
      ddpm: x' = c1 (x - c2 eps) + c3 z,    c1 = 1 / sqrt(a_t), c2 = (1 - a_t) / sqrt(1 - abar_t), c3 = sqrt(beta_tilde)."""
    sched = InferenceSchedule(inference_schedule=inference_schedule, T=diffusion.T, inference_T=steps)
    rows = []
    for i in range(steps):
        t1, t2 = (int(v) for v in sched(i))
        atbar, atbar_target = diffusion.alphabar[t1], diffusion.alphabar[t2]
        if trick == "ddpm" or t1 <= 1:
            at = diffusion.alpha[t1]
            atbar_prev = diffusion.alphabar[t1 - 1]
            beta_tilde = diffusion.beta[t1 - 1] * (1 - atbar_prev) / (1 - atbar)
            rows.append([t1, 1.0, (1 / np.sqrt(at)).item(), ((1 - at) / np.sqrt(1 - atbar)).item(), np.sqrt(beta_tilde).item(), t2])
        elif trick == "ddim":
            rows.append([t1, 0.0, np.sqrt(atbar_target / atbar).item(), np.sqrt(1 - atbar).item(), np.sqrt(1 - atbar_target).item(), t2])
        else:
            raise ValueError("Unknown inference trick {}".format(trick))
    return np.asarray(rows, dtype=np.float64)
