"""CPU tests of the f(x) restatement (oracle/ns_transformer_oracle.py, PARITY UNPINNED -- see its header) and of the holders that carry
the same parameter names into the native library."""
import types

import pytest
import torch


def _cfg(**kw):
    from oracle import ns_transformer_oracle as O
    return O, O.config(**kw)


@pytest.mark.parametrize("latent", [False, True])
def test_holder_and_oracle_agree_on_names_and_shapes(latent):
    O, cfg = _cfg(dataset_nf=2, windows=24, pred_len=10, d_model=32, n_heads=2, d_ff=64, e_layers=2, d_layers=1)
    sd = O.random_state(cfg, 0, latent=latent)
    if latent:
        from mpv_amd.models.Diffusion_model.TMDM import tmdm_ns_transformer as H
        ns = types.SimpleNamespace(**vars(cfg), enc_in=2, dec_in=2, c_out=2)
        m = H.Model(ns)
    else:
        from mpv_amd.models.Diffusion_model.NsDiff import mu_backbone as H
        m = H.Model(cfg)
    own = m.state_dict()
    assert set(own) == set(sd)
    assert all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)
    m.load_state_dict(sd, strict=True)
    fields, tensors = m.native_weight_list()
    assert fields["latent"] == int(latent) and fields["pe_rows"] == 24
    assert len(tensors) == 4 + 16 * 2 + 2 + 26 * 1 + 4 + 2 * (2 + 2 * 2) + (8 if latent else 0)      # the count mpv_nst_load checks
    assert torch.equal(own["enc_embedding.position_embedding.pe"], O.position_table(5000, 32))


def test_restated_forward_properties():
    O, cfg = _cfg(dataset_nf=2, windows=24, pred_len=10, d_model=32, n_heads=2, d_ff=64, e_layers=1, d_layers=1)
    sd = O.random_state(cfg, 1, latent=True)
    torch.manual_seed(0)
    x = torch.randn(3, 24, 2).cumsum(1)
    pred, full = O.forward(sd, cfg, x)
    assert tuple(pred.shape) == (3, 10, 2) and tuple(full.shape) == (3, 22, 2) and torch.equal(full[:, -10:], pred)
    assert not torch.allclose(O.forward(sd, cfg, x, latent=True)[0], pred)
    # rows are independent, and the decoder's self-attention is causal: with the cross-attention and the encoder fixed, an output
    # position only depends on decoder inputs at or before it -> perturbing the LAST history sample (decoder position label_len-1)
    # leaves nothing to check there, so check batch independence and determinism instead
    again = O.forward(sd, cfg, torch.cat([x[1:2], x[0:1]]))[0]
    assert torch.allclose(again[0], pred[1], atol=1e-6) and torch.allclose(again[1], pred[0], atol=1e-6)
    # ds_attention with tau = 1, delta = 0 is plain softmax attention
    q, k, v = torch.randn(2, 5, 2, 4), torch.randn(2, 7, 2, 4), torch.randn(2, 7, 2, 4)
    want = torch.softmax(torch.einsum("blhe,bshe->bhls", q, k) / 2.0, -1)
    assert torch.allclose(O.ds_attention(q, k, v, torch.ones(2, 1), torch.zeros(2, 7), False), torch.einsum("bhls,bshd->blhd", want, v), atol=1e-6)
    causal = O.ds_attention(q[:, :5], k[:, :5], v[:, :5], None, None, True)
    assert torch.allclose(causal[:, 0], v[:, 0], atol=1e-6)                        # the first query sees only the first key


def test_models_build_the_backbone_only_when_configured():
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.models.Diffusion_model.NsDiff import mu_backbone
    base = dict(device="cpu", scaler_type="StandardScaler", rolling_length=10, diffusion_schedule="linear", diffusion_steps=5, beta_start=1e-4,
                beta_end=0.02, n_z_samples=4, parallel_sample=2, windows=24, pred_len=10, dataset_nf=2, load_pretrain=False)
    fx = dict(d_model=32, n_heads=2, d_ff=64, e_layers=1, d_layers=1, activation="gelu", p_hidden_dims=[16, 16], p_hidden_layers=2)
    m = diffusion_models("NsDiff", dict(base, **fx), train_model_select="NsDiff_model")
    assert isinstance(m.cond_pred_model, mu_backbone.Model) and any(k.startswith("cond_pred_model.decoder.") for k in m.state_dict())
    assert diffusion_models("NsDiff", dict(base), train_model_select="NsDiff_model").cond_pred_model is None
    f_only = diffusion_models("NsDiff", dict(base, **fx), train_model_select="pretrain_f")
    assert isinstance(f_only.cond_pred_model, mu_backbone.Model) and getattr(f_only, "model", None) is None
    v = diffusion_models("NsDiff_model_variants", dict(base, **fx), train_model_select="cond_mean")
    assert isinstance(v.cond_pred_model, mu_backbone.Model)
    assert diffusion_models("NsDiff_model_variants", dict(base, **fx), train_model_select="cond_var").cond_pred_model is None
    from mpv_amd import _native
    with pytest.raises(_native.MpvError):                                          # CPU: f(x) is native too, no fallback
        m.evaluation_step(torch.zeros(2, 24, 2))
    with pytest.raises(NotImplementedError, match="output_attention"):
        diffusion_models("NsDiff", dict(base, **fx, output_attention=True), train_model_select="NsDiff_model")
