"""Noise schedule + epsilon-network container of TMDM (host side; reference TMDM/TMDM.py:22-98).

Same table attribute names, same `diffussion_model.*` (sic) parameter names.  `enc_embedding` of the
reference (`DataEmbedding`, third-party) is not instantiated: with `cat_x and cat_y_pred` its output never
reaches the epsilon network (tmdm_model.py:39-41), so it is dead compute on this path; its checkpoint
entries are accepted and ignored by `TMDM_model.load_state_dict`.
"""
from __future__ import annotations

import torch.nn as nn

from .tmdm_diffusion_utils import tmdm_coef_rows, tmdm_tables
from .tmdm_model import ConditionalGuidedModel


class TMDM(nn.Module):
    def __init__(self, configs, device):
        super().__init__()
        self.args = configs
        self.device = device
        self.num_timesteps = int(configs.timesteps)
        schedule = getattr(configs, "beta_schedule", "linear")          # tmdm.yml defaults (:33-39)
        start = getattr(configs, "beta_start", 0.0001)
        end = getattr(configs, "beta_end", 0.02)
        tb = tmdm_tables(self.num_timesteps, schedule, start, end)
        for name, val in tb.items():
            setattr(self, name, val)
        self.coef_rows = tmdm_coef_rows(tb)
        self.diffussion_model = ConditionalGuidedModel(self.num_timesteps, int(configs.enc_in))

    def native_weights(self):
        m = self.diffussion_model
        lins = (m.lin1, m.lin2, m.lin3)
        return dict(lin_w=[l.lin.weight for l in lins], lin_b=[l.lin.bias for l in lins],
                    embed=[l.embed.weight for l in lins], eps_w=m.lin4.weight, eps_b=m.lin4.bias,
                    coef=self.coef_rows)

    def forward(self, *args, **kwargs):
        return self.diffussion_model(*args, **kwargs)
