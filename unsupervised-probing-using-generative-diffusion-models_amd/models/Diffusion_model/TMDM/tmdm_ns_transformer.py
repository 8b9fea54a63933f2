"""TMDM's prior network: the ns-Transformer of NsDiff with a latent bottleneck between encoder and decoder.

Mirrors reference TMDM/tmdm_ns_transformer.py:41-172 in constructor argument (`configs`) and parameter names (the NsDiff backbone's
names plus `z_mean`, `z_logvar`, `z_out`).  In eval mode the reference's forward is deterministic: z = z_mean(enc_out), the
decoder attends to z_out(z), and `z_logvar` only feeds the KL term of the training loss (:155-162), so the native call
(csrc/nst.hip, `latent = 1`) returns the two prediction tensors and `None` for KL_z / z_sample.  Same parity status as the NsDiff
backbone: the third-party layers are restated, PARITY UNPINNED.
"""
from __future__ import annotations

from ..NsDiff import mu_backbone as _base


class Model(_base.Model):
    latent = True

    def __init__(self, configs, native=None):
        if getattr(configs, "c_out", configs.enc_in) != configs.enc_in or getattr(configs, "dec_in", configs.enc_in) != configs.enc_in:
            raise ValueError("the prior is de-normalised with the input statistics: enc_in, dec_in and c_out must agree")
        super().__init__(configs, native=native, enc_in=configs.enc_in)

    def forward(self, x_enc, x_mark_enc=None, x_dec=None, x_mark_dec=None, enc_self_mask=None, dec_self_mask=None, dec_enc_mask=None):
        """(batch_x, None, dec_inp, None) -> (dec_out[:, -pred_len:], dec_out, None, None)   (reference :136-172, eval mode)."""
        if x_mark_enc is not None or x_mark_dec is not None:
            raise NotImplementedError("time-mark embeddings are not on this path (the adapter passes None, tmdm_adapter.py:123)")
        if enc_self_mask is not None or dec_self_mask is not None or dec_enc_mask is not None:
            raise NotImplementedError("attention masks other than the decoder's causal mask are not part of this path")
        dec_out = self._run(x_enc)
        return dec_out[:, -self.pred_len:, :], dec_out, None, None
