"""sigma head g(x): weight holder + native call.

Mirrors the reference's `SigmaEstimation` (models/Diffusion_model/NsDiff/g_backbone.py:19-72):
same constructor signature and the same `mlp.{0,2,3,5,6}` parameter names, so
`cond_pred_model_g.*` checkpoint entries load strictly.  `forward` runs the trailing
variance (NsDiff/sigma.py:34-72) and the 3-layer MLP in the HIP kernel of
csrc/sigma_head.hip through the C ABI; there is no torch implementation here.
"""
from __future__ import annotations

import torch
import torch.nn as nn


class SigmaEstimation(nn.Module):
    def __init__(self, seq_len, pred_len, enc_in, hidden_size=512, kernel_size=24, native=None):
        super().__init__()
        if not (1 <= kernel_size < seq_len):
            raise ValueError("rolling_length must satisfy 1 <= rolling_length < windows "
                             "(got rolling_length={}, windows={})".format(kernel_size, seq_len))
        self.pred_len, self.seq_len, self.enc_in = pred_len, seq_len, enc_in
        self.hidden_size, self.kernel_size = hidden_size, kernel_size
        self.padding = kernel_size // 2
        self.mlp = nn.Sequential(
            nn.Linear(seq_len - kernel_size, hidden_size), nn.ReLU(), nn.LayerNorm([enc_in, hidden_size]),
            nn.Linear(hidden_size, hidden_size), nn.ReLU(), nn.LayerNorm([enc_in, hidden_size]),
            nn.Linear(hidden_size, pred_len),
        )
        # plain attribute (not a submodule / not in state_dict): the owning model's NativeHandle
        object.__setattr__(self, "_native", native)

    def native_weights(self):
        m = self.mlp
        return [m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[3].weight, m[3].bias,
                m[5].weight, m[5].bias, m[6].weight, m[6].bias]

    def forward(self, x_enc: torch.Tensor) -> torch.Tensor:
        """x_enc [B, T, F] -> sigma^2 predictions [B, pred_len, F] on x_enc's device."""
        if x_enc.dim() != 3:
            raise ValueError("x_enc must be a 3D tensor with shape (B, T, N)")
        if x_enc.shape[1] != self.seq_len or x_enc.shape[2] != self.enc_in:
            raise ValueError("sigma head was built for windows [B, {}, {}], got {}".format(
                self.seq_len, self.enc_in, tuple(x_enc.shape)))
        if self._native is None:
            raise RuntimeError("SigmaEstimation is not attached to a native handle")
        ctx = self._native.context(self)
        series = x_enc.to(device=ctx.device, dtype=torch.float32).contiguous()
        return ctx.sigma_head(series, 1, 0, 1)[0]
