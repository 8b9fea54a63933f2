"""Weight holders for TMDM's pointwise conditional MLP (reference TMDM/tmdm_model.py:6-64): same parameter
names, embedding tables with the reference's T+1 rows, so its state_dict entries load strictly."""
from __future__ import annotations

import torch.nn as nn

HIDDEN = 128


class ConditionalLinear(nn.Module):
    def __init__(self, num_in, num_out, n_steps):
        super().__init__()
        self.num_out = num_out
        self.lin = nn.Linear(num_in, num_out)
        self.embed = nn.Embedding(n_steps, num_out)
        self.embed.weight.data.uniform_()


class ConditionalGuidedModel(nn.Module):
    """`cat_x and cat_y_pred` configuration only (the shipped tmdm.yml): input cat(y_t, y_0_hat)."""

    def __init__(self, timesteps: int, enc_in: int):
        super().__init__()
        n_steps = timesteps + 1                                       # tmdm_model.py:25
        self.lin1 = ConditionalLinear(enc_in * 2, HIDDEN, n_steps)
        self.lin2 = ConditionalLinear(HIDDEN, HIDDEN, n_steps)
        self.lin3 = ConditionalLinear(HIDDEN, HIDDEN, n_steps)
        self.lin4 = nn.Linear(HIDDEN, enc_in)

    def forward(self, *args, **kwargs):
        raise RuntimeError("the epsilon network runs inside the fused HIP chain kernel; "
                           "call the owning model's evaluation_step")
