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


# ---- Model_spatial: f(x) of NsDiff_spatial (reference NsDiff/mu_backbone.py:186-340) --------------------------------------------------------
SPATIAL = dict(dataset_nf=2, windows=24, pred_len=12, d_model=32, n_heads=2, d_ff=64, e_layers=1, d_layers=1, fT_h=4, spatial_layers=2)


def dense_images(sd, S, D, fT_h):
    """The dense [fT_h D] x [S D] and [S D] x [fT_h D] images of the k = S + 1 convolutions as csrc/nst.hip's mpv_nst_load lays them out
    (Conv2d: k = ti - to + pad; ConvTranspose2d, weight [in][out]: k = t - to + pad), restated here index for index."""
    p, K = fT_h // 2, S + 1
    wd, wu = sd["downsampling.weight"][:, :, 0, :], sd["upsampling.weight"][:, :, 0, :]
    down = torch.zeros(fT_h * D, S * D)
    up = torch.zeros(S * D, fT_h * D)
    for to in range(fT_h):
        for ti in range(S):
            k = ti - to + p
            if 0 <= k < K:
                down[to * D:(to + 1) * D, ti * D:(ti + 1) * D] = wd[:, :, k]
    for t in range(S):
        for to in range(fT_h):
            k = t - to + p
            if 0 <= k < K:
                up[t * D:(t + 1) * D, to * D:(to + 1) * D] = wu[:, :, k].t()            # [ci][co] -> rows co, columns ci
    return down, up


def test_dense_images_of_the_long_convolutions_equal_the_convolutions():
    """The library runs Model_spatial's downsampling / upsampling convolutions as ONE dense product each over [rows][S D] / [rows][fT_h D];
    this checks that layout against torch's conv2d / conv_transpose2d on the reference's tensor shapes (mu_backbone.py:312-325)."""
    import torch.nn.functional as F
    O, cfg = _cfg(**SPATIAL)
    sd = O.random_spatial_state(cfg, 3)
    S, D, Tn = cfg.seq_len, cfg.d_model, cfg.fT_h
    down, up = dense_images(sd, S, D, Tn)
    g = torch.Generator().manual_seed(1)
    enc = torch.randn(5, S, D, generator=g)
    h = F.conv2d(enc.unsqueeze(2).transpose(1, 3), sd["downsampling.weight"], sd["downsampling.bias"], padding=(0, Tn // 2)).transpose(1, 3).squeeze(2)
    assert tuple(h.shape) == (5, Tn, D)
    got = enc.reshape(5, -1) @ down.t() + sd["downsampling.bias"].repeat(Tn)
    assert torch.allclose(got.reshape(5, Tn, D), h, atol=2e-5)
    feat = torch.randn(5, Tn, D, generator=g)
    u = F.conv_transpose2d(feat.unsqueeze(2).transpose(1, 3), sd["upsampling.weight"], sd["upsampling.bias"], padding=(0, Tn // 2)).transpose(1, 3).squeeze(2)
    assert tuple(u.shape) == (5, S, D)
    got = feat.reshape(5, -1) @ up.t() + sd["upsampling.bias"].repeat(S)
    assert torch.allclose(got.reshape(5, S, D), u, atol=2e-5)


def test_spatial_holder_names_tensor_list_and_model_wiring():
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.models.Diffusion_model.NsDiff import mu_backbone as H
    O, cfg = _cfg(**SPATIAL)
    sd = O.random_spatial_state(cfg, 5)
    m = H.Model_spatial(cfg)
    own = m.state_dict()
    assert set(own) == set(sd) and all(tuple(own[k].shape) == tuple(sd[k].shape) for k in sd)
    assert tuple(own["downsampling.weight"].shape) == (32, 32, 1, 25) and tuple(own["spatial_encoder.1.gnn.lin_skip.weight"].shape) == (128, 128)
    m.load_state_dict(sd, strict=True)
    fields, tensors = m.native_weight_list()
    assert fields["fT_h"] == 4 and fields["spatial_layers"] == 2 and fields["latent"] == 0
    assert len(tensors) == 4 + 16 * 1 + 2 + 26 * 1 + 4 + 2 * (2 + 2 * 2) + 4 + 8 * 2                  # the count mpv_nst_load checks
    # the restated stage: rows of different graphs do not mix, an isolated node only sees its own skip path
    edge = torch.tensor([[0, 1, 1, 2], [1, 0, 2, 1]])
    torch.manual_seed(0)
    x = torch.randn(4, 24, 2).cumsum(1)                                                                # node 3 has no edges
    pred, full = O.forward_spatial(sd, cfg, x, edge)
    assert tuple(pred.shape) == (4, 12, 2) and torch.equal(full[:, -12:], pred)
    x2 = x.clone()
    x2[0] += 1.0
    pred2, _ = O.forward_spatial(sd, cfg, x2, edge)
    assert torch.allclose(pred2[3], pred[3], atol=1e-6) and not torch.allclose(pred2[1], pred[1], atol=1e-4)    # 0 -> 1 is an edge, nothing reaches 3
    base = dict(device="cpu", scaler_type="StandardScaler", rolling_length=8, diffusion_schedule="linear", diffusion_steps=5, beta_start=1e-4,
                beta_end=0.02, n_z_samples=4, parallel_sample=2, windows=24, pred_len=12, dataset_nf=2, load_pretrain=False, d_h=4, Td_h=4,
                channel_multipliers=[1], n_blocks=1, gnn_name="ResGatedGraphConv", gnn_param={})
    fx = dict(d_model=32, n_heads=2, d_ff=64, e_layers=1, d_layers=1, activation="gelu", p_hidden_dims=[64, 64], p_hidden_layers=2, fT_h=4, spatial_layers=2)
    try:
        full_model = diffusion_models("NsDiff_spatial", dict(base, **fx), train_model_select="NsDiff_model")
    except KeyError as exc:                     # the sampler network's own hyper-parameters are another test's business
        pytest.skip("NsDiff_net_spatial needs %s" % exc)
    assert isinstance(full_model.cond_pred_model, H.Model_spatial)
    assert any(k.startswith("cond_pred_model.spatial_encoder.0.gnn.lin_key.") for k in full_model.state_dict())
    f_only = diffusion_models("NsDiff_spatial", dict(base, **fx), train_model_select="pretrain_f")
    assert isinstance(f_only.cond_pred_model, H.Model_spatial) and getattr(f_only, "model", None) is None
    with pytest.raises(NotImplementedError, match="fT_h"):
        diffusion_models("NsDiff_spatial", dict(base), train_model_select="pretrain_f")
    with pytest.raises(NotImplementedError, match="no native kernel"):
        H.Model_spatial(types.SimpleNamespace(**vars(cfg), f_gnn_name="PNAConv"))
    gcn = H.Model_spatial(types.SimpleNamespace(**vars(cfg), f_gnn_name="GCNConv"))         # round 4: SAGE / Graph / GCN / GAT layers are native too
    f2, t2 = gcn.native_weight_list()
    assert f2["spatial_gnn"] == 12 and len(t2) == len(tensors) and sum(t is None for t in t2) == 6 * 2 and "spatial_encoder.0.gnn.lin.weight" in gcn.state_dict()
