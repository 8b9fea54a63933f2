"""TMDM sampler.

CPU part: pins oracle/tmdm_oracle.py and the host tables against vectors recorded from the reference's
own TMDM code (tests/golden/make_golden_tmdm.py).  GPU part (-m gpu): mpv_tmdm_sample through the C ABI
and through TMDM_model against the same vectors.  Tolerance 1e-4 relative (north_star), fp32."""
import numpy as np
import pytest
import torch

from oracle import philox as PH
from oracle import tmdm_oracle as TO

T = torch.from_numpy
RTOL = 1e-4
CASES = [("f1t50", 1, 50), ("f2t10", 2, 10), ("f3t100", 3, 100)]


def close(a, b, rtol=2e-6, atol=1e-7):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def rel_err(got, want):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    want = want.detach().cpu().double().numpy() if isinstance(want, torch.Tensor) else np.asarray(want, dtype=np.float64)
    return float(np.abs(got - want).max() / max(np.abs(want).max(), 1e-30))


def state(golden_tmdm, tag):
    pre = "sd_%s_" % tag
    return {k[len(pre):]: T(golden_tmdm[k]) for k in golden_tmdm.files if k.startswith(pre)}


# ---- CPU: oracle and host tables vs the reference ---------------------------------------------
@pytest.mark.parametrize("steps,kind", [(10, "linear"), (50, "linear"), (100, "linear"), (12, "cosine"), (8, "quad")])
def test_tables_bit_exact(golden_tmdm, steps, kind):
    from mpv_amd.models.Diffusion_model.TMDM.tmdm_diffusion_utils import tmdm_tables
    for tb in (TO.schedule_tables(steps, kind), tmdm_tables(steps, kind, 1e-4, 0.02)):
        assert np.array_equal(tb["alphas"].numpy(), golden_tmdm["sched_%s_%d_alphas" % (kind, steps)])
        assert np.array_equal(tb["one_minus_alphas_bar_sqrt"].numpy(), golden_tmdm["sched_%s_%d_s" % (kind, steps)])


@pytest.mark.parametrize("tag,nf,steps", CASES)
def test_oracle_matches_reference(golden_tmdm, tag, nf, steps):
    sd, tb = state(golden_tmdm, tag), TO.schedule_tables(steps)
    assert TO.random_state_dict(nf, steps, seed={"f1t50": 31, "f2t10": 32, "f3t100": 33}[tag]).keys() == sd.keys()
    y, y0 = T(golden_tmdm["in_%s_y" % tag]), T(golden_tmdm["in_%s_y0" % tag])
    for t in range(steps):
        close(TO.eps_network(sd, y, y0, t), golden_tmdm["eps_%s" % tag][t], rtol=1e-5, atol=1e-6)
    for t in (steps - 1, steps // 2, 1):
        z = T(golden_tmdm["step_%s_t%d_z" % (tag, t)])
        close(TO.reverse_step(sd, tb, y, y0, y0, t, z), golden_tmdm["step_%s_t%d_out" % (tag, t)], rtol=1e-5, atol=1e-6)
    close(TO.final_step(sd, tb, y, y0, y0), golden_tmdm["step_%s_final" % tag], rtol=1e-5, atol=1e-6)
    out = TO.sample_loop(sd, tb, y0, y0, steps, _replay(golden_tmdm["loop_%s_noise" % tag]))
    assert rel_err(out, golden_tmdm["loop_%s_out" % tag]) < 2e-5


def _replay(arr):
    it = iter(T(np.ascontiguousarray(arr)))
    return lambda shape: next(it).reshape(shape)


@pytest.mark.parametrize("tag,nf,steps", CASES)
def test_oracle_evaluation_step(golden_tmdm, tag, nf, steps):
    sd, tb = state(golden_tmdm, tag), TO.schedule_tables(steps)
    B, label_len, pred_len, n_z, par = (int(v) for v in golden_tmdm["eval_%s_dims" % tag])
    prior = T(golden_tmdm["eval_%s_prior" % tag])
    draws = T(golden_tmdm["eval_%s_noise" % tag]).reshape(n_z // par, steps, B * par, label_len + pred_len, nf)
    out = TO.evaluation_step(sd, tb, prior, steps, pred_len, n_z, par, _replay(draws.reshape(-1, *draws.shape[2:]).numpy()))
    assert tuple(out.shape) == (B, pred_len, nf, n_z)
    assert rel_err(out, golden_tmdm["eval_%s_out" % tag]) < 2e-5
    with pytest.raises(ValueError):
        TO.evaluation_step(sd, tb, prior, steps, pred_len, 7, 3, lambda s: torch.zeros(s))


@pytest.mark.parametrize("steps", [10, 50])
def test_coef_rows_are_the_reference_scalars(steps):
    from mpv_amd.models.Diffusion_model.TMDM.tmdm_diffusion_utils import tmdm_coef_rows, tmdm_tables
    tb = tmdm_tables(steps, "linear", 1e-4, 0.02)
    rows = tmdm_coef_rows(tb)
    assert tuple(rows.shape) == (steps, 20)
    for t in range(1, steps):
        want = torch.stack(TO.step_scalars(tb, t))
        assert torch.equal(rows[t, 3:7], want)
    s = tb["one_minus_alphas_bar_sqrt"]
    r = (1 - s.square()).sqrt()
    assert torch.equal(rows[:, 0], 1 / r) and torch.equal(rows[:, 1], 1 - r) and torch.equal(rows[:, 2], s)


def _model(golden_tmdm, tag, nf, steps, device="cpu", **kw):
    from mpv_amd.models.models import diffusion_models
    B, label_len, pred_len, n_z, par = (int(v) for v in golden_tmdm["eval_%s_dims" % tag])
    param = dict(device=device, dataset_nf=nf, windows=2 * label_len, pred_len=pred_len, diffusion_steps=steps,
                 n_z_samples=n_z, parallel_sample=par, scaler_type="standard")
    param.update(kw)
    m = diffusion_models("TMDM", param)
    sd = state(golden_tmdm, tag)
    sd["scaler_mean"], sd["scaler_std"] = torch.zeros(nf), torch.ones(nf)
    # a reference checkpoint also carries the prior network (loaded into the holder) and the unused data embedding (dropped)
    torch.manual_seed(5)
    for k, v in m.cond_pred_model.state_dict().items():
        sd["cond_pred_model." + k] = v.clone() if k.endswith(".pe") else torch.randn_like(v) * 0.1
    sd["model.enc_embedding.position_embedding.pe"] = torch.zeros(1, 8, 4)
    m.load_state_dict(sd)
    assert torch.equal(m.cond_pred_model.z_out[2].weight, sd["cond_pred_model.z_out.2.weight"])
    return m.to(device)


def test_host_model_mirrors_reference_interface(golden_tmdm):
    m = _model(golden_tmdm, "f2t10", 2, 10)
    assert m.label_len == 4 and m.seq_len == 8 and m.model.num_timesteps == 10
    assert m.model.diffussion_model.lin1.embed.weight.shape == (11, 128)              # T+1 rows
    x = torch.arange(3 * 8 * 2, dtype=torch.float32).reshape(3, 8, 2)
    assert torch.equal(m.scaler_inverse_transform(m.scaler_transform(x)), x)
    assert m._decoder_input(x).shape == (3, 4 + 8, 2)
    from mpv_amd import _native
    with pytest.raises(_native.MpvError):
        m.evaluation_step(x)                                                          # CPU: the prior is native too, no fallback
    prior, m.cond_pred_model = m.cond_pred_model, None
    with pytest.raises(NotImplementedError):
        m.evaluation_step(x)                                                          # prior removed and no y_0_hat
    m.cond_pred_model = prior
    with pytest.raises(_native.MpvError):
        m.evaluation_step(x, y_0_hat=torch.zeros(3, 12, 2))                           # CPU: no fallback
    m.n_z_samples, m.parallel_sample = 7, 3
    with pytest.raises(ValueError):
        m.evaluation_step(x, y_0_hat=torch.zeros(3, 12, 2))
    with pytest.raises(RuntimeError):
        m.model(x)


# ---- GPU: the HIP path ------------------------------------------------------------------------
def dev():
    return torch.device("cuda", 0)


def make_ctx(sd, nf, steps):
    from mpv_amd import _native
    from mpv_amd.models.Diffusion_model.TMDM.tmdm_diffusion_utils import tmdm_coef_rows, tmdm_tables
    ctx = _native.Context(dev())
    pre = TO.EPS_NET
    ctx.load_tmdm(steps, nf, [sd[pre + "lin%d.lin.weight" % i] for i in (1, 2, 3)],
                  [sd[pre + "lin%d.lin.bias" % i] for i in (1, 2, 3)],
                  [sd[pre + "lin%d.embed.weight" % i] for i in (1, 2, 3)],
                  sd[pre + "lin4.weight"], sd[pre + "lin4.bias"], tmdm_coef_rows(tmdm_tables(steps, "linear", 1e-4, 0.02)))
    return ctx


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag,nf,steps", CASES)
def test_gpu_chain_matches_reference_loop(golden_tmdm, tag, nf, steps, gemm):
    ctx = make_ctx(state(golden_tmdm, tag), nf, steps)
    y0 = T(golden_tmdm["in_%s_y0" % tag]).to(dev())[None].contiguous()               # [1, 6, 9, F]
    noise = T(np.ascontiguousarray(golden_tmdm["loop_%s_noise" % tag])).to(dev())     # [T, 6, 9, F]
    out = ctx.sample_tmdm(y0, 1, 1, noise=noise, gemm=gemm)[0, :, :, :, 0]
    err = rel_err(out, golden_tmdm["loop_%s_out" % tag])
    print("tmdm chain %s [%s] rel err %.3e" % (tag, gemm, err))
    assert err < RTOL


@pytest.mark.gpu
@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag,nf,steps", CASES)
def test_gpu_evaluation_step_replays_reference_noise_stream(golden_tmdm, tag, nf, steps, gemm):
    """TMDM_model.evaluation_step under the reference's torch.manual_seed reproduces the reference's outs."""
    m = _model(golden_tmdm, tag, nf, steps, device="cuda:0", noise_source="torch_cpu", gemm_mode=gemm)
    B, label_len, pred_len, n_z, par = (int(v) for v in golden_tmdm["eval_%s_dims" % tag])
    prior = T(golden_tmdm["eval_%s_prior" % tag])
    batch = torch.zeros(B, 2 * label_len, nf)
    torch.manual_seed(99)
    outs, batch_y = m.evaluation_step(batch, y_0_hat=prior)
    assert batch_y is None and tuple(outs.shape) == (B, pred_len, nf, n_z) and outs.device.type == "cpu"
    err = rel_err(outs, golden_tmdm["eval_%s_out" % tag])
    print("tmdm evaluation_step %s [%s] rel err %.3e" % (tag, gemm, err))
    assert err < RTOL
    torch.manual_seed(99)
    outs2, _ = m.evaluation_step(batch, y_0_hat=prior[:, -pred_len:, :])                # pre-cut prior: same result
    assert torch.equal(outs, outs2)


@pytest.mark.gpu
def test_gpu_philox_mode_and_ragged_shapes(golden_tmdm):
    """Philox-mode launch == oracle fed the normals the device dumped; ragged tile counts; empty batch."""
    nf, steps = 2, 10
    sd = state(golden_tmdm, "f2t10")
    ctx = make_ctx(sd, nf, steps)
    from mpv_amd.models.Diffusion_model.NsDiff.NsDiff_net import nsdiff_tables, posterior_coef_rows
    from oracle import nsdiff_oracle as NO
    nsd = NO.random_state_dict(nf, steps, 8, 8, 4, seed=1)
    pre = "model.diffussion_model."
    ctx.load_nsdiff(steps, nf, [nsd[pre + "lin%d.lin.weight" % i] for i in (1, 2, 3)],
                    [nsd[pre + "lin%d.lin.bias" % i] for i in (1, 2, 3)],
                    [nsd[pre + "lin%d.embed.weight" % i] for i in (1, 2, 3)], nsd[pre + "lin4.weight"], nsd[pre + "lin4.bias"],
                    nsd[pre + "sigma_lin.weight"], nsd[pre + "sigma_lin.bias"],
                    posterior_coef_rows(nsdiff_tables(steps, "linear", 1e-4, 0.02)))       # philox_normals needs (T, F)
    tb = TO.schedule_tables(steps)
    for (W, B, L, S) in ((2, 3, 7, 5), (1, 1, 3, 1), (3, 5, 11, 4)):
        n = W * B * L * S
        z = ctx.philox_normals(41, 3, n).cpu()
        assert np.abs(z.numpy() - PH.normals(41, 3, n, steps, nf)).max() < 5e-5
        g = torch.Generator().manual_seed(W)
        prior = torch.randn(W, B, L, nf, generator=g) * 0.3
        rows = prior.reshape(-1, 1, nf).repeat_interleave(S, dim=0)
        draws = [z[:, k, :].reshape(n, 1, nf) for k in range(steps)]
        want = TO.sample_loop(sd, tb, rows, rows, steps, _replay(torch.stack(draws).numpy()))
        want = want.reshape(W, B, L, S, nf).permute(0, 1, 2, 4, 3)
        for gemm in ("f16x3", "f32"):
            got = ctx.sample_tmdm(prior.to(dev()), S, S, seed=41, offset=3, gemm=gemm).cpu()
            assert rel_err(got, want) < RTOL, gemm
    out = ctx.sample_tmdm(torch.zeros(0, 3, 4, nf, device=dev()), 4, 4, seed=1)
    assert tuple(out.shape) == (0, 3, 4, nf, 4)
    from mpv_amd._native import MpvError
    with pytest.raises(MpvError):
        ctx.sample_tmdm(torch.zeros(1, 1, 4, nf, device=dev()), 6, 4, seed=1)


@pytest.mark.gpu
def test_gpu_guarded_path_with_large_activations(golden_tmdm):
    """Scaled-up weights push the un-normalised activations past the f16 range of the fast path: the step re-runs
    through the exact routine (normalised B operands) and still matches the oracle; both GEMM modes."""
    nf, steps = 2, 10
    sd = dict(state(golden_tmdm, "f2t10"))
    for i in (2, 3):
        sd[TO.EPS_NET + "lin%d.lin.weight" % i] = sd[TO.EPS_NET + "lin%d.lin.weight" % i] * 400.0
    ctx = make_ctx(sd, nf, steps)
    g = torch.Generator().manual_seed(4)
    prior = torch.randn(1, 3, 5, nf, generator=g) * 0.3
    noise = torch.randn(1, 1, steps, 3, 5, nf, generator=g)
    rows = prior[0]
    want = TO.sample_loop(sd, TO.schedule_tables(steps), rows, rows, steps, _replay(noise[0, 0].numpy()))
    assert bool(torch.isfinite(want).all())
    for gemm in ("f16x3", "f32"):
        got = ctx.sample_tmdm(prior.to(dev()), 1, 1, noise=noise.to(dev()), gemm=gemm)[0, :, :, :, 0].cpu()
        assert rel_err(got, want) < RTOL, gemm


@pytest.mark.gpu
def test_gpu_harness_window_loop_with_attached_prior(golden_tmdm, tmp_path):
    """run_evaluation_cache / summarize_pred_future_list drive a TMDM model whose prior is an attached torch module:
    per-window evaluation_step on the native sampler, MPV from the device reduction == float64 torch on the cache."""
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    nf, steps = 2, 10
    m = _model(golden_tmdm, "f2t10", nf, steps, device="cuda:0")

    class Prior(torch.nn.Module):                      # reference call signature (tmdm_adapter.py:122-123)
        def forward(self, x, x_mark, dec_inp, dec_mark):
            return None, dec_inp * 0.0 + x.mean(dim=1, keepdim=True), None, None

    m.cond_pred_model = Prior().to("cuda:0")
    g = torch.Generator().manual_seed(3)
    series = torch.randn(4, 60, nf, generator=g)
    wins, tp = H.build_sliding_windows(series, np.arange(60) * 0.1, m.windows, 10)
    preds = H.run_evaluation_cache(m, list(wins), m.pred_len, tmp_path / "c.pt", torch.device("cuda", 0), force_recompute=True)
    assert len(preds) == len(wins) and tuple(preds[0].shape) == (4, m.pred_len, nf, m.n_z_samples)
    pm, ews = H.summarize_pred_future_list(preds, model=m)
    want = torch.stack([p.double().var(dim=-1, unbiased=False).mean() for p in preds])
    assert rel_err(torch.tensor(np.array(ews, dtype=np.float64)), want) < 1e-5
    assert len(H._load_tensor_list(tmp_path / "c.pt")) == len(wins)


@pytest.mark.gpu
def test_gpu_harness_batches_windows_through_the_native_prior(golden_tmdm, tmp_path):
    """TMDM in the harness with its own (native) prior network: one prior call and one sampler launch for a block of windows gives what
    the reference's window-by-window loop of evaluation_step gives (shared reference noise stream)."""
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    nf, steps = 2, 10
    m = _model(golden_tmdm, "f2t10", nf, steps, device="cuda:0", noise_source="torch_cpu")
    g = torch.Generator().manual_seed(3)
    series = torch.randn(4, 60, nf, generator=g)
    wins, tp = H.build_sliding_windows(series, np.arange(60) * 0.1, m.windows, 10)
    torch.manual_seed(8)
    preds = H.run_evaluation_cache(m, list(wins), m.pred_len, tmp_path / "c.pt", torch.device("cuda", 0), force_recompute=True)
    torch.manual_seed(8)
    loop = [m.evaluation_step(m.scaler_transform(torch.as_tensor(w).float().to("cuda:0")))[0] for w in wins]
    assert len(preds) == len(loop) == len(wins)
    for a, b in zip(preds, loop):
        assert tuple(a.shape) == (4, m.pred_len, nf, m.n_z_samples) and rel_err(a, b) < 1e-5
