"""Pin oracle/diffusionts_oracle.py (no HIP path yet) against vectors recorded from the
reference's own DiffusionTS package (tests/golden/make_golden_diffusionts.py)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import diffusionts_oracle as D

GOLD = Path(__file__).resolve().parent / "golden" / "diffusionts_cases.npz"
T = torch.from_numpy


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def case(gold, tag):
    pre = "sd_%s_" % tag
    sd = {k[len(pre):]: T(gold[k]) for k in gold.files if k.startswith(pre)}
    w, o, nf, d, le, ld, nh, ts, steps, nz, par = (int(v) for v in gold["cfg_%s" % tag])
    cfg = D.DTSConfig(windows=w, pred_len=o, nf=nf, d_model=d, n_layer_enc=le, n_layer_dec=ld, n_heads=nh, timesteps=ts,
                      sampling_timesteps=steps, n_z_samples=nz, parallel_sample=par)
    return sd, cfg


def replay(flat, sizes):
    chunks = list(torch.split(T(np.ascontiguousarray(flat)), [int(s) for s in sizes]))
    it = iter(chunks)
    return lambda shape: next(it).reshape(shape)


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_schedule_buffers_bit_exact(gold, tag):
    sd, cfg = case(gold, tag)
    tb = D.schedule_tables(cfg.timesteps)
    for name, val in tb.items():
        assert torch.equal(val, sd["model." + name]), name


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_transformer_and_model_predictions(gold, tag):
    sd, cfg = case(gold, tag)
    x = T(gold["x_%s" % tag])
    with torch.no_grad():
        for t in (0, cfg.timesteps // 2, cfg.timesteps - 1):
            trend, season = D.transformer(sd, cfg, x, torch.full((3,), t, dtype=torch.long))
            np.testing.assert_allclose(trend.numpy(), gold["tf_%s_t%d_trend" % (tag, t)], rtol=2e-5, atol=2e-6)
            np.testing.assert_allclose(season.numpy(), gold["tf_%s_t%d_season" % (tag, t)], rtol=2e-5, atol=2e-6)
        tb = D.schedule_tables(cfg.timesteps)
        pn, x0 = D.model_predictions(sd, cfg, tb, x, torch.full((3,), cfg.timesteps // 3, dtype=torch.long), clip_x_start=True)
    np.testing.assert_allclose(x0.numpy(), gold["mp_%s_x0" % tag], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(pn.numpy(), gold["mp_%s_noise" % tag], rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_evaluation_step_with_the_reference_noise_stream(gold, tag):
    """DDIM + Langevin infill + the sample-major / node-major tiling quirk, on the reference's recorded draws."""
    sd, cfg = case(gold, tag)
    used = []
    out = D.evaluation_step(sd, cfg, T(gold["batch_%s" % tag]), replay(gold["eval_%s_noise" % tag], gold["eval_%s_ndraws" % tag]), record=used)
    want = gold["eval_%s_out" % tag]
    assert tuple(out.shape) == want.shape and len(used) == len(gold["eval_%s_ndraws" % tag])
    err = float(np.abs(out.detach().numpy() - want).max() / np.abs(want).max())
    print("DiffusionTS oracle evaluation_step [%s]: max rel err %.3e over %d draws" % (tag, err, len(used)))
    assert err < 2e-4
    with pytest.raises(ValueError):
        D.evaluation_step(sd, D.DTSConfig(cfg.windows, cfg.pred_len, cfg.nf, n_z_samples=5, parallel_sample=2), T(gold["batch_%s" % tag]),
                          lambda s: torch.zeros(s))


def test_eta_sampler_matches_reference_and_host_plan_carries_sigma():
    """eta != 0 (reference DiffusionTS.py:294-300, 392-395: stochastic DDIM step, sigma-weighted Langevin terms): the oracle against
    evaluation_step recorded from the reference (tests/golden/make_golden_diffusionts_eta.py, the `small` model), and the host plan's sigma
    column against the reference's formula."""
    from conftest import GOLDEN
    from mpv_amd.models.models import diffusion_models
    z, ze = np.load(GOLDEN / "diffusionts_cases.npz"), np.load(GOLDEN / "diffusionts_eta_cases.npz")
    keys = ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec", "n_heads", "timesteps", "diffusion_steps", "n_z_samples", "parallel_sample")
    par = dict(zip(keys, [int(v) for v in z["cfg_small"]]))
    sd = {k[len("sd_small_"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd_small_")}
    L = par["windows"] + par["pred_len"]
    for tag, eta in (("eta05", 0.5), ("eta1", 1.0)):
        cfg = D.DTSConfig(par["windows"], par["pred_len"], par["dataset_nf"], par["d_model"], par["n_layer_enc"], par["n_layer_dec"], par["n_heads"],
                          par["timesteps"], par["diffusion_steps"], n_z_samples=4, parallel_sample=2, eta=eta)
        it = iter(torch.from_numpy(ze["eval_%s_noise" % tag]).reshape(-1, 4, L, par["dataset_nf"]).unbind(0))
        out = D.evaluation_step(sd, cfg, torch.from_numpy(ze["batch"]), lambda shape: next(it))
        assert torch.equal(out, torch.from_numpy(ze["eval_%s_out" % tag]))
        m = diffusion_models("DiffusionTS", dict(par, eta=eta, device="cpu", scaler_type="StandardScaler"))
        plan = m._plan()
        assert tuple(plan.shape) == (par["diffusion_steps"], 11)
        tb = D.schedule_tables(par["timesteps"])
        for row in plan.tolist():
            t, tn = int(row[0]), int(row[1])
            if tn < 0:
                assert row[10] == 0.0
                continue
            a, an = tb["alphas_cumprod"][t], tb["alphas_cumprod"][tn]
            sigma = eta * ((1 - a / an) * (1 - an) / (1 - a)).sqrt()
            assert row[10] == float(sigma) and row[5] == float((1 - an - sigma ** 2).sqrt())
    m0 = diffusion_models("DiffusionTS", dict(par, device="cpu", scaler_type="StandardScaler"))
    assert float(m0._plan()[:, 10].abs().max()) == 0.0


def test_full_plan_fixture_small_model_and_c3full_header():
    """tests/golden/diffusionts_c3full.npz (make_golden_diffusionts_c3full.py, round 4): the REAL 100 / 100-step plan -- 100 model_predictions + 128
    Langevin forward / backward passes per chain with the K in {0, 1, 2, 3} thresholds of DiffusionTS.py:359-407 -- run by the reference itself.
    `t100`: a small model with every draw recorded; the oracle replays it.  `c3full`: 25 nodes x 32 samples = 800 sequences of the BASELINE
    configs[2] model in one chunk (85 minutes of 6 CPU threads); only its header is checked here, the -m gpu test compares the device run."""
    from conftest import GOLDEN
    z = np.load(GOLDEN / "diffusionts_c3full.npz")
    w, o, nf, d, le, ld, nh, ts, steps, nz, par = (int(v) for v in z["cfg_t100"])
    assert ts == 100 and steps == 100
    sd = {k[len("sd_t100_"):]: T(z[k]) for k in z.files if k.startswith("sd_t100_")}
    cfg = D.DTSConfig(windows=w, pred_len=o, nf=nf, d_model=d, n_layer_enc=le, n_layer_dec=ld, n_heads=nh, timesteps=ts, sampling_timesteps=steps,
                      n_z_samples=nz, parallel_sample=par)
    L, rows = w + o, par * 2
    draws = T(z["eval_t100_noise"]).reshape(-1, rows, L, nf)
    assert draws.shape[0] == (nz // par) * 327                               # 1 + sum over the 99 pairs of (2 + K): 1 + 198 + 128 per chunk
    it = iter(draws.unbind(0))
    used = []
    out = D.evaluation_step(sd, cfg, T(z["batch_t100"]), lambda shape: next(it).reshape(shape), record=used)
    want = z["eval_t100_out"]
    err = float(np.abs(out.detach().numpy() - want).max() / np.abs(want).max())
    print("DiffusionTS oracle on the 100 / 100-step plan (small model): max rel err %.3e over %d draws" % (err, len(used)))
    assert len(used) == draws.shape[0] and err < 2e-4
    nodes, n_z, n_steps = (int(v) for v in z["c3full_cfg"][:3])
    assert (nodes, n_z, n_steps) == (25, 32, 100) and tuple(z["c3full_out"].shape) == (25, 100, 1, 32) and tuple(z["c3full_hist"].shape) == (25, 100, 1)
    x = z["c3full_out"].astype(np.float64)
    assert abs(float(x.var(axis=-1).mean()) - float(z["c3full_mpv"])) < 1e-12 and abs(float(x.mean()) - float(z["c3full_pred_mean"])) < 1e-12
    assert np.isfinite(x).all() and np.abs(x).max() <= 1.0 + 1e-6              # the last step returns the clamped x0 estimate
