"""GPU parity tests (-m gpu) of f(x), the ns-Transformer mean backbone, and of TMDM's prior,
through the C ABI (mpv_nst_load / mpv_nst_forward) against oracle/ns_transformer_oracle.py.

PARITY UNPINNED: the oracle restates the torch-timeseries==0.1.10 layers the reference imports (package absent, no network), so
these tests pin the HIP kernels to the restatement -- wiring, Projector, statistics and de-normalisation are the reference's own
lines (NsDiff/mu_backbone.py:12-183) -- not to a run of the reference.  Tolerance 1e-4 elementwise in both gemm modes: "f32" runs the Linear
layers and the attention on the exact-fp32 matrix pipe, "f16x3" (default) runs both as two-term f16 splits with fp32 accumulation
(nst_gemm_h3_kernel, and since round 4 nst_attn_h_kernel at head widths 16 / 32 / 64, causal and not); what differs from the restatement is
summation order in LayerNorm / softmax / the k-split of the products."""
import pytest
import torch

from test_gpu_parity import dev, rel_err

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def oracle():
    from oracle import ns_transformer_oracle as O
    return O


def holder(cfg, sd, latent=False):
    """A model object that owns a backbone holder + native handle (the TMDM adapter or the NsDiff model); returns (model, backbone)."""
    from mpv_amd.models.models import diffusion_models
    common = dict(device=dev(), dataset_nf=cfg.dataset_nf, windows=cfg.seq_len, pred_len=cfg.pred_len, d_model=cfg.d_model, n_heads=cfg.n_heads,
                  d_ff=cfg.d_ff, e_layers=cfg.e_layers, d_layers=cfg.d_layers, activation=cfg.activation, p_hidden_dims=list(cfg.p_hidden_dims),
                  p_hidden_layers=cfg.p_hidden_layers, n_z_samples=4, parallel_sample=2)
    if latent:
        m = diffusion_models("TMDM", dict(common, diffusion_steps=5, scaler_type=None))
    else:
        m = diffusion_models("NsDiff", dict(common, diffusion_steps=5, scaler_type="StandardScaler", rolling_length=max(2, cfg.seq_len // 4),
                                            diffusion_schedule="linear", beta_start=1e-4, beta_end=0.02, load_pretrain=False),
                             train_model_select="NsDiff_model")
    m.cond_pred_model.load_state_dict(sd, strict=True)                      # the oracle's names ARE the holder's names
    with torch.no_grad():
        m.scaler_mean.zero_()
        m.scaler_std.fill_(1.0)
    return m.to(dev()).eval(), m.cond_pred_model


CASES = {
    # tag: (config kwargs, batch, latent)
    "small_e32": (dict(dataset_nf=2, windows=40, pred_len=24, d_model=64, n_heads=2, d_ff=32, e_layers=2, d_layers=1), 5, False),
    "relu_p1": (dict(dataset_nf=3, windows=24, pred_len=10, d_model=32, n_heads=2, d_ff=64, e_layers=1, d_layers=2, activation="relu",
                     p_hidden_dims=(48,), p_hidden_layers=1), 7, False),
    "tmdm_default": (dict(dataset_nf=2, windows=32, pred_len=16, d_model=64, n_heads=4, d_ff=128, e_layers=2, d_layers=1), 6, True),
    "slbp": (dict(dataset_nf=2, windows=200, pred_len=200), 3, False),          # shipped NsDiff YAMLs: d_model 512, 8 heads, 4 + 2 layers
    "c2": (dict(dataset_nf=1, windows=100, pred_len=100), 12, False),
}


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", list(CASES))
def test_forward_matches_the_restatement(tag, gemm):
    O = oracle()
    kw, B, latent = CASES[tag]
    cfg = O.config(**kw)
    sd = O.random_state(cfg, seed=len(tag), latent=latent)
    m, f = holder(cfg, sd, latent)
    f.gemm_mode = gemm
    torch.manual_seed(3)
    x = torch.randn(B, cfg.seq_len, cfg.dataset_nf).cumsum(1) * 0.3 + torch.randn(B, 1, cfg.dataset_nf)
    want_pred, want_full = O.forward(sd, cfg, x, latent=latent)
    out = f(x.to(dev()), None, None, None) if latent else f(x.to(dev()), None)
    assert tuple(out[0].shape) == tuple(want_pred.shape) and tuple(out[1].shape) == tuple(want_full.shape)
    print(tag, gemm, "rel_err", rel_err(out[1], want_full))
    assert rel_err(out[1], want_full) < RTOL, rel_err(out[1], want_full)
    assert rel_err(out[0], want_pred) < RTOL
    if not latent:
        inter = O.intermediates(sd, cfg, x)
        ctx = m.native_context()
        assert rel_err(ctx.nst_debug("x_norm", B), inter["x_norm"]) < 1e-5
        assert rel_err(ctx.nst_debug("stats", B), torch.cat([inter["mean"], inter["std"]], 1)) < 1e-5
        assert rel_err(ctx.nst_debug("tau", B), inter["tau"].reshape(-1)) < 1e-5
        assert rel_err(ctx.nst_debug("delta", B), inter["delta"]) < 1e-5
        assert rel_err(ctx.nst_debug("enc_out", B), inter["enc_out"]) < RTOL


def test_windows_of_a_resident_series_with_the_scaler_fused_and_row_passes():
    O = oracle()
    cfg = O.config(dataset_nf=2, windows=40, pred_len=24, d_model=64, n_heads=2, d_ff=32, e_layers=2, d_layers=1)
    sd = O.random_state(cfg, seed=9)
    m, f = holder(cfg, sd)
    torch.manual_seed(4)
    V, L, first, step, W = 6, 120, 3, 7, 11
    series = torch.randn(V, L, 2).cumsum(1) * 0.2 + 5.0
    mean, std = series.mean((0, 1)), series.std((0, 1))
    ctx = m.native_context()
    got = ctx.nst_forward(series.to(dev()), W, first, step, mean, std)
    assert tuple(got.shape) == (W, V, cfg.pred_len, 2)
    wins = torch.stack([series[:, first + w * step: first + w * step + cfg.seq_len] for w in range(W)])
    want = O.forward(sd, cfg, ((wins - mean) / std).reshape(W * V, cfg.seq_len, 2))[0].reshape(W, V, cfg.pred_len, 2)
    assert rel_err(got, want) < RTOL
    split = ctx.nst_forward(series.to(dev()), W, first, step, mean, std, rows_per_pass=13)       # 66 rows in passes of 13
    assert torch.equal(split, got)
    full = ctx.nst_forward(series.to(dev()), W, first, step, mean, std, full=True)
    assert tuple(full.shape) == (W, V, cfg.label_len + cfg.pred_len, 2) and torch.equal(full[:, :, -cfg.pred_len:], got)
    from mpv_amd import _native
    with pytest.raises(_native.MpvError, match="past the end"):
        ctx.nst_forward(series.to(dev()), W, first, step + 2, mean, std)


def test_full_nsdiff_samples_around_the_native_mean():
    """task_model NsDiff / train_model_select NsDiff_model: f(x), g(x), sampler and reduction in one native call; the same
    samples as feeding the oracle's y_0_hat through the `y0hat=` input with the same Philox seed."""
    O = oracle()
    cfg = O.config(dataset_nf=2, windows=40, pred_len=24, d_model=64, n_heads=2, d_ff=32, e_layers=2, d_layers=1)
    sd = O.random_state(cfg, seed=2)
    m, f = holder(cfg, sd)
    torch.manual_seed(8)
    V, W = 5, 4
    series = torch.randn(V, 80, 2).cumsum(1) * 0.2
    with torch.no_grad():
        m.scaler_mean.copy_(series.mean((0, 1)))
        m.scaler_std.copy_(series.std((0, 1)))
    samples, gx, mpv, pm = m.sample_windows(series.to(dev()), W, 0, 10, scale_input=True, seed=77, summary=True)
    wins = torch.stack([series[:, 10 * w: 10 * w + 40] for w in range(W)]).reshape(W * V, 40, 2)
    y0 = O.forward(sd, cfg, (wins - series.mean((0, 1))) / series.std((0, 1)))[0].reshape(W, V, 24, 2)
    ref, _, mpv_ref, pm_ref = m.sample_windows(series.to(dev()), W, 0, 10, scale_input=True, seed=77, summary=True, y0hat=y0)
    assert rel_err(samples, ref) < 2e-4
    assert rel_err(mpv, mpv_ref) < 1e-4 and rel_err(pm, pm_ref) < 1e-4
    outs, _ = m.evaluation_step((wins[:V] - series.mean((0, 1))) / series.std((0, 1)))                   # evaluation_step takes scaled windows
    assert tuple(outs.shape) == (V, 24, 2, 4) and bool(torch.isfinite(outs).all())


def test_tmdm_evaluation_step_uses_the_native_prior():
    O = oracle()
    cfg = O.config(dataset_nf=2, windows=32, pred_len=16, d_model=64, n_heads=4, d_ff=128, e_layers=2, d_layers=1)
    sd = O.random_state(cfg, seed=6, latent=True)
    m, f = holder(cfg, sd, latent=True)
    torch.manual_seed(1)
    x = torch.randn(4, 32, 2).cumsum(1) * 0.3
    prior = O.forward(sd, cfg, x, latent=True)[1]
    torch.manual_seed(21)
    got, _ = m.evaluation_step(x)
    torch.manual_seed(21)
    want, _ = m.evaluation_step(x, y_0_hat=prior)
    assert tuple(got.shape) == (4, 16, 2, 4)
    assert rel_err(got, want) < 2e-4


def test_load_rejects_unsupported_shapes():
    O = oracle()
    from mpv_amd import _native
    cfg = O.config(dataset_nf=1, windows=16, pred_len=8, d_model=48, n_heads=1, d_ff=32, e_layers=1, d_layers=1)      # head width 48
    m, f = holder(cfg, O.random_state(cfg, 0))
    with pytest.raises(_native.MpvError, match="head width"):
        f(torch.zeros(2, 16, 1, device=dev()), None)


def test_uncertainty_ews_on_a_full_nsdiff_model_directory(tmp_path, monkeypatch):
    """A `task_model: NsDiff` / `train_model_select: NsDiff_model` directory in the reference's on-disk layout (checkpoint dict with
    `cond_pred_model.*`, YAML with the backbone's keys) goes through `uncertainty_ews` end to end: strict load, f(x) -> g(x) -> sampler ->
    reduction on device.  The curve equals a second run through the `y0hat=` input fed with the restatement's mean (same draws)."""
    import numpy as np
    from helpers import write_model_dir
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from oracle import nsdiff_oracle as NO
    O = oracle()
    cfg = O.config(dataset_nf=1, windows=40, pred_len=24, d_model=64, n_heads=2, d_ff=32, e_layers=2, d_layers=1)
    sd = NO.random_state_dict(1, 10, 40, 24, 16, seed=11)                                  # sampler + sigma head + scaler
    sd["scaler_mean"], sd["scaler_std"] = torch.tensor([0.8]), torch.tensor([0.4])
    fx = O.random_state(cfg, seed=5)
    sd.update({"cond_pred_model." + k: v for k, v in fx.items()})
    net = dict(device="cuda", scaler_type="StandardScaler", task_model="NsDiff", rolling_length=16, diffusion_schedule="linear", diffusion_steps=10,
               beta_start=1e-4, beta_end=0.02, n_z_samples=8, parallel_sample=4, windows=40, pred_len=24, dataset_nf=1, load_pretrain=False,
               d_model=64, n_heads=2, d_ff=32, e_layers=2, d_layers=1, activation="gelu", p_hidden_dims=[64, 64], p_hidden_layers=2,
               output_attention=False, noise_source="torch_cpu")
    mdir = tmp_path / "model"
    write_model_dir(mdir, sd, net, select="NsDiff_model")
    g = torch.Generator().manual_seed(9)
    series = (torch.randn(200, 6, generator=g).cumsum(0) * 0.05 + 0.8)                      # [ticks, V]
    from test_gpu_parity import _reseed_after
    _reseed_after(monkeypatch, H, "load_model_from_dir", 17)                                  # (constructing the model draws from the stream)
    res = H.uncertainty_ews(model_save_file=mdir, torch_time_series=series, time_data=torch.arange(200) * 0.1, dynamic_type="biomass",
                            sampling_t=0.1, sample_window_step=40, device=dev(), save_nsdiff_g=False)
    assert len(res["ews"]) == 5 and np.isfinite(np.array(res["ews"], dtype=np.float64)).all()
    model, _ = H.load_model_from_dir(mdir, device=torch.device(dev()))
    assert type(model.cond_pred_model).__name__ == "Model" and torch.equal(model.cond_pred_model.decoder.projection.weight.cpu(),
                                                                               fx["decoder.projection.weight"])
    # the same five windows with the mean supplied from the CPU restatement instead of the native backbone
    full = series.t().unsqueeze(-1)                                                          # [V, ticks, 1]
    wins = torch.stack([full[:, 40 * w: 40 * w + 40] for w in range(5)])
    scaled = (wins - sd["scaler_mean"]) / sd["scaler_std"]
    y0 = O.forward(fx, cfg, scaled.reshape(30, 40, 1))[0].reshape(5, 6, 24, 1)
    torch.manual_seed(17)
    noise = model.host_noise(5, 6)
    _, _, mpv, pm = model.sample_windows(full.to(dev()), 5, 0, 40, scale_input=True, y0hat=y0, noise=noise, summary=True)
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), mpv.cpu().double().numpy(), rtol=2e-4)
    np.testing.assert_allclose(np.array(res["pred_mean"], dtype=np.float64), pm.cpu().double().numpy(), rtol=2e-4, atol=1e-6)
