"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes) and through
the reference-shaped model API, against the reference's golden vectors and the CPU oracle
on the same inputs and the same noise.  Tolerance: north_star's 1e-4 relative on fp32."""
import numpy as np
import pytest
import torch

from oracle import nsdiff_oracle as O
from oracle import philox as PH

pytestmark = pytest.mark.gpu

T = torch.from_numpy
RTOL = 1e-4


def dev():
    return torch.device("cuda", 0)


def rel_err(got, want):
    """ELEMENTWISE relative error with a floor of a tenth of the tensor's RMS: `rel_err < 1e-4` means
    |got - want| <= 1e-4 |want| + 1e-5 rms(want) for every element (small-magnitude samples are constrained to 1e-5 of the
    typical magnitude, not just to 1e-4 of the largest one)."""
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, dtype=np.float64)
    want = want.detach().cpu().double().numpy() if isinstance(want, torch.Tensor) else np.asarray(want, dtype=np.float64)
    if want.size == 0:
        return 0.0
    floor = 0.1 * float(np.sqrt(np.mean(np.square(want)))) + 1e-30
    return float((np.abs(got - want) / (np.abs(want) + floor)).max())


def make_ctx(sd, nf, steps, sigma=None):
    from mpv_amd import _native
    from mpv_amd.models.Diffusion_model.NsDiff.NsDiff_net import nsdiff_tables, posterior_coef_rows
    ctx = _native.Context(dev())
    pre = O.EPS_NET
    ctx.load_nsdiff(steps, nf, [sd[pre + "lin%d.lin.weight" % i] for i in (1, 2, 3)],
                    [sd[pre + "lin%d.lin.bias" % i] for i in (1, 2, 3)],
                    [sd[pre + "lin%d.embed.weight" % i] for i in (1, 2, 3)],
                    sd[pre + "lin4.weight"], sd[pre + "lin4.bias"], sd[pre + "sigma_lin.weight"], sd[pre + "sigma_lin.bias"],
                    posterior_coef_rows(nsdiff_tables(steps, "linear", 1e-4, 0.02)))
    if sigma is not None:
        seq, k, O_ = sigma
        g = O.G_NET
        ctx.load_sigma(seq, k, O_, nf, 512, [sd[g + n] for n in ("0.weight", "0.bias", "2.weight", "2.bias", "3.weight",
                                                                   "3.bias", "5.weight", "5.bias", "6.weight", "6.bias")])
    return ctx


def model_sd(tag, ckpt_wofx):
    if tag == "wofx20":
        return ckpt_wofx, 2, 20
    if tag == "seed50":
        return O.random_state_dict(1, 50, 100, 100, 50, seed=123), 1, 50
    return O.random_state_dict(2, 10, 200, 200, 100, seed=321), 2, 10


def test_library_loads_and_reports_device():
    from mpv_amd import _native
    lib = _native.load_library()
    assert lib.mpv_abi_version() == _native.ABI_VERSION
    ctx = _native.Context(dev())
    with pytest.raises(_native.MpvError):
        ctx.sample(torch.ones(1, 1, 4, 1, device=dev()), None, 4, 4)          # nothing loaded yet


@pytest.mark.parametrize("gemm", ["f16x3", "f32", "f16x2", "f16"])
@pytest.mark.parametrize("tag", ["seed50", "wofx20", "seed10"])
def test_chain_matches_reference_loop(golden_cases, ckpt_wofx, tag, gemm):
    """p_sample_loop golden (reference run, recorded noise) through mpv_nsdiff_sample."""
    sd, nf, steps = model_sd(tag, ckpt_wofx)
    ctx = make_ctx(sd, nf, steps)
    y0 = T(golden_cases["step_%s_y0" % tag]).to(dev())[None].contiguous()      # [1, 6, 5, F]
    gx = T(golden_cases["step_%s_gx" % tag]).to(dev())[None].contiguous()
    noise = T(golden_cases["loop_%s_noise" % tag]).to(dev()).contiguous()       # [T, 6, 5, F]
    out = ctx.sample(gx, y0, 1, 1, noise=noise, gemm=gemm)[0, :, :, :, 0]
    err = rel_err(out, golden_cases["loop_%s_out" % tag])
    print("chain %s [%s] rel err %.3e" % (tag, gemm, err))
    assert err < RTOL


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
def test_chain_nan_on_negative_discriminant_like_reference(ckpt_wofx, gemm):
    """gx = 0 makes lambda_0 > 0, lambda_2 = 0 and -lambda_1 +- |lambda_1| -> the reference's own
    arithmetic (sqrt(noise) of a tiny negative, 0/0 ...) must be reproduced, not 'fixed': compare with
    the oracle elementwise including where it is NaN."""
    sd, nf, steps = O.random_state_dict(1, 10, 100, 100, 50, seed=5), 1, 10
    ctx = make_ctx(sd, nf, steps)
    tb = O.schedule_tables(steps)
    gx = torch.tensor([0.0, 1e-12, 1e-3, 4.0]).reshape(1, 4, 1, 1)
    g = torch.Generator().manual_seed(3)
    noise = torch.randn(steps, 4, 1, 1, generator=g)
    want = O.sample_loop(sd, tb, torch.zeros(4, 1, 1), gx[0], torch.zeros(4, 1, 1), steps, O.replay_noise(list(noise)))
    got = ctx.sample(gx.to(dev()), None, 1, 1, noise=noise.to(dev()), gemm=gemm)[0, :, :, :, 0].cpu()
    assert torch.equal(torch.isnan(got), torch.isnan(want))
    ok = ~torch.isnan(want)
    assert rel_err(got[ok], want[ok]) < RTOL


def test_sigma_head_matches_reference(golden_cases, ckpt_wofx):
    ctx = make_ctx(ckpt_wofx, 2, 20, sigma=(200, 100, 200))
    x = T(golden_cases["ghead_wofx_x"]).to(dev())
    got = ctx.sigma_head(x, 1, 0, 1)[0]
    err = rel_err(got, golden_cases["ghead_wofx_y"])
    print("sigma head (real ckpt) rel err %.3e" % err)
    assert err < 1e-5
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    ctx = make_ctx(sd, 1, 50, sigma=(100, 50, 100))
    got = ctx.sigma_head(T(golden_cases["ghead_seed_x"]).to(dev()), 1, 0, 1)[0]
    assert rel_err(got, golden_cases["ghead_seed_y"]) < 1e-5


def test_sigma_head_fused_windows_and_scaler(golden_cases):
    """Sliding-window gather + scaler fused into the kernel == per-window oracle (a1-a5)."""
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    sd["scaler_mean"] = T(golden_cases["eval_seed_scaler_mean"])
    sd["scaler_std"] = T(golden_cases["eval_seed_scaler_std"])
    ctx = make_ctx(sd, 1, 50, sigma=(100, 50, 100))
    series = T(np.ascontiguousarray(golden_cases["net_series"]))                            # [3, 260, 1]
    gx = ctx.sigma_head(series.to(dev()), 3, 0, 80, sd["scaler_mean"].to(dev()), sd["scaler_std"].to(dev()))
    assert rel_err(gx, golden_cases["net_gx"]) < 1e-5                   # reference run_nsdiff_g_cache
    # ragged tail: 1 window starting at an odd offset, V not a multiple of the kernel's group size
    gx1 = ctx.sigma_head(series.to(dev()), 1, 37, 5, sd["scaler_mean"].to(dev()), sd["scaler_std"].to(dev()), add_eps=1e-7)
    want = O.sigma_head(sd, O.scale(series[:, 37:137], sd["scaler_mean"], sd["scaler_std"]), 50, 100) + 1e-7
    assert rel_err(gx1[0], want) < 1e-5
    from mpv_amd._native import MpvError
    with pytest.raises(MpvError):
        ctx.sigma_head(series.to(dev()), 3, 0, 81)                     # last window past the end


@pytest.mark.parametrize("S", [1, 4, 32, 33, 100])
def test_mpv_reduce_matches_torch(S):
    from mpv_amd import _native
    ctx = _native.Context(dev())
    g = torch.Generator().manual_seed(S)
    x = torch.randn(5, 37, 2, S, generator=g) * 3 + 1
    std, mean = torch.tensor([2.0, 0.5]), torch.tensor([-1.0, 4.0])
    mpv, pm, feat = ctx.reduce(x.to(dev()), std.to(dev()), mean.to(dev()), per_feature=True)
    for w in range(5):
        m, v = O.mpv_network(x[w].double(), mean.double(), std.double())
        assert abs(mpv[w].item() - v) <= 2e-6 * abs(v) + 1e-12
        assert abs(pm[w].item() - m) <= 2e-6 * abs(m) + 1e-7
    mpv2, _, feat2 = ctx.reduce(x.to(dev()), per_feature=True)            # no inverse scaling (SLBP summary)
    want = x.double().var(dim=-1, unbiased=False).mean(dim=1)              # [W, F]
    assert rel_err(feat2, want) < 2e-6


def _variant_model(sd, select, nf, steps, windows, pred_len, rolling, n_z, par, noise_source="philox", gemm="f16x3"):
    from mpv_amd.models.models import diffusion_models
    net_param = dict(device=dev(), scaler_type="StandardScaler", task_model="NsDiff_model_variants",
                     rolling_length=rolling, diffusion_schedule="linear", diffusion_steps=steps, beta_start=1e-4,
                     beta_end=0.02, n_z_samples=n_z, parallel_sample=par, windows=windows, pred_len=pred_len,
                     dataset_nf=nf, noise_source=noise_source, gemm_mode=gemm)
    model = diffusion_models("NsDiff_model_variants", net_param, train_model_select=select).to(dev())
    own = model.state_dict()
    model.load_state_dict({k: sd[k] for k in own}, strict=True)
    return model.eval()


@pytest.mark.parametrize("gemm", ["f16x3", "f32", "f16x2", "f16"])
def test_evaluation_step_real_checkpoint(golden_cases, ckpt_wofx, gemm):
    """C1 shape, real wo_fx checkpoint, two chunks: the reference's recorded noise injected."""
    model = _variant_model(ckpt_wofx, "cond_var", 2, 20, 200, 200, 100, 4, 2, gemm=gemm)
    outs, by = model.evaluation_step(T(golden_cases["eval_wofx_batch"]), noise=T(golden_cases["eval_wofx_noise"]))
    assert by is None and outs.device.type == "cpu" and tuple(outs.shape) == (1, 200, 2, 4)
    err = rel_err(outs, golden_cases["eval_wofx_out"])
    print("evaluation_step wo_fx [%s] rel err %.3e" % (gemm, err))
    assert err < RTOL


@pytest.mark.parametrize("gemm", ["f16x3", "f32", "f16x2", "f16"])
def test_evaluation_step_second_real_checkpoint_against_oracle(ckpt_sim, gemm):
    """The other shipped sampler checkpoint (real_data/model/simulation: F=1, windows=pred_len=200, rolling 50, T=20,
    trained weights) against the pinned oracle on shared noise, 3 rows x 4 samples in 2 chunks; MPV too."""
    model = _variant_model(ckpt_sim, "cond_var", 1, 20, 200, 200, 50, 4, 2, gemm=gemm)
    cfg = O.NsDiffConfig(200, 200, 1, 50, 20, n_z_samples=4, parallel_sample=2)
    g = torch.Generator().manual_seed(31)
    batch = torch.cumsum(torch.randn(3, 200, 1, generator=g) * 0.1, dim=1)
    rec = []
    torch.manual_seed(5)
    want = O.evaluation_step(ckpt_sim, cfg, batch, record=rec)
    outs, _ = model.evaluation_step(batch, noise=torch.cat([z.reshape(-1) for z in rec]))
    err = rel_err(outs, want)
    mpv_w = want.var(dim=-1, unbiased=False).mean()
    mpv_g = outs.var(dim=-1, unbiased=False).mean()
    print("evaluation_step simulation ckpt [%s] rel err %.3e, MPV rel err %.3e" % (gemm, err, abs(mpv_g - mpv_w) / mpv_w))
    assert err < RTOL and abs(mpv_g - mpv_w) / mpv_w < RTOL


@pytest.mark.parametrize("gemm", ["f16x3", "f32", "f16x2", "f16"])
def test_evaluation_step_seeded_and_torch_cpu_noise_stream(golden_cases, gemm):
    """Reduced C2 shape; noise_source='torch_cpu' + the reference's seed == the reference run."""
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    model = _variant_model(sd, "cond_var", 1, 50, 100, 100, 50, 8, 4, noise_source="torch_cpu", gemm=gemm)
    torch.manual_seed(12)
    outs, _ = model.evaluation_step(T(golden_cases["eval_seed_batch"]))
    err = rel_err(outs, golden_cases["eval_seed_out"])
    print("evaluation_step seeded [%s] rel err %.3e" % (gemm, err))
    assert err < RTOL
    m, v = O.mpv_network(outs)
    mw, vw = O.mpv_network(T(golden_cases["eval_seed_out"]))
    assert abs(v - vw) <= RTOL * abs(vw)


def test_evaluation_step_gaussian_variant(golden_cases):
    sd = O.random_state_dict(2, 10, 200, 200, 100, seed=321)
    model = _variant_model(sd, "Guassian", 2, 10, 200, 200, 100, 4, 4)
    outs, _ = model.evaluation_step(T(golden_cases["eval_gauss_batch"]), noise=T(golden_cases["eval_gauss_noise"]))
    assert rel_err(outs, golden_cases["eval_gauss_out"]) < RTOL


@pytest.mark.parametrize("nf,steps", [(2, 10), (1, 50), (1, 7)])
def test_philox_mode_replays_through_oracle(nf, steps):
    """Device Philox normals == numpy restatement, and a Philox-mode launch == the oracle fed
    the normals the device dumped."""
    sd = O.random_state_dict(nf, steps, 200, 200, 100, seed=321)
    ctx = make_ctx(sd, nf, steps)
    W, B, Ol, S = 2, 3, 7, 5
    n_chains = W * B * Ol * S
    z = ctx.philox_normals(99, 5, n_chains).cpu()                              # [n, T, F]
    zn = PH.normals(99, 5, n_chains, steps, nf)
    assert np.abs(z.numpy() - zn).max() < 5e-5
    assert abs(float(z.mean())) < 0.05 and abs(float(z.std()) - 1) < 0.05
    g = torch.Generator().manual_seed(1)
    gx = torch.rand(W, B, Ol, nf, generator=g) + 0.05
    y0 = torch.randn(W, B, Ol, nf, generator=g) * 0.1
    got = ctx.sample(gx.to(dev()), y0.to(dev()), S, S, seed=99, offset=5).cpu()   # [W,B,O,F,S]
    # oracle rows = chains in (w,b,o,s) order, horizon axis of length 1
    rows_g = gx.reshape(-1, 1, nf).repeat_interleave(S, dim=0)
    rows_y = y0.reshape(-1, 1, nf).repeat_interleave(S, dim=0)
    draws = [z[:, k, :].reshape(n_chains, 1, nf) for k in range(steps)]
    want = O.sample_loop(sd, O.schedule_tables(steps), rows_y, rows_g, rows_y, steps, O.replay_noise(draws))
    want = want.reshape(W, B, Ol, S, nf).permute(0, 1, 2, 4, 3)
    assert rel_err(got, want) < RTOL
    again = ctx.sample(gx.to(dev()), y0.to(dev()), S, S, seed=99, offset=5).cpu()
    assert torch.equal(got, again)                                            # counter-based: bitwise repeatable
    other = ctx.sample(gx.to(dev()), y0.to(dev()), S, S, seed=100, offset=5).cpu()
    assert not torch.equal(got, other)


@pytest.mark.parametrize("nf,gemms", [(3, ("f16x3", "f32", "f16x2", "f16")), (4, ("f32",))])
def test_three_and_four_feature_series_against_oracle(nf, gemms):
    """F = 3 (split-f16 and exact kernels) and F = 4 (exact kernel only; the split-f16 one refuses it) on seeded weights,
    injected noise, y0hat supplied."""
    steps = 12
    sd = O.random_state_dict(nf, steps, 40, 24, 16, seed=77)
    ctx = make_ctx(sd, nf, steps)
    g = torch.Generator().manual_seed(2)
    B, Ol = 5, 7
    gx = torch.rand(1, B, Ol, nf, generator=g) + 0.1
    y0 = torch.randn(1, B, Ol, nf, generator=g) * 0.2
    noise = torch.randn(steps, B, Ol, nf, generator=g)
    want = O.sample_loop(sd, O.schedule_tables(steps), y0[0], gx[0], y0[0], steps, O.replay_noise(list(noise)))
    for gemm in gemms:
        got = ctx.sample(gx.to(dev()), y0.to(dev()), 1, 1, noise=noise.to(dev()), gemm=gemm)[0, :, :, :, 0].cpu()
        assert rel_err(got, want) < RTOL, (nf, gemm)
    if nf == 4:
        from mpv_amd._native import MpvError
        with pytest.raises(MpvError, match="nf 1..3"):
            ctx.sample(gx.to(dev()), y0.to(dev()), 1, 1, noise=noise.to(dev()), gemm="f16x3")


def test_fast_and_guarded_activation_paths_are_both_exercised(ckpt_wofx):
    """Seeded-init weights prove |pre-activation| < 4 (fast epilogue); scaled-up weights force the guarded one,
    including pre-activations beyond torch's softplus threshold (x > 20)."""
    sd = O.random_state_dict(1, 10, 100, 100, 50, seed=5)
    ctx = make_ctx(sd, 1, 10)
    big, huge = dict(sd), dict(sd)
    for i in (2, 3):
        big[O.EPS_NET + "lin%d.lin.weight" % i] = sd[O.EPS_NET + "lin%d.lin.weight" % i] * 60.0
        huge[O.EPS_NET + "lin%d.lin.weight" % i] = sd[O.EPS_NET + "lin%d.lin.weight" % i] * 4000.0    # 2^x' overflows fp32
    ctx_big, ctx_huge = make_ctx(big, 1, 10), make_ctx(huge, 1, 10)
    g = torch.Generator().manual_seed(9)
    gx = (torch.rand(1, 4, 6, 1, generator=g) + 0.1)
    noise = torch.randn(10, 4, 6, 1, generator=g)
    tb = O.schedule_tables(10)
    for c, w in ((ctx, sd), (ctx_big, big), (ctx_huge, huge)):
        want = O.sample_loop(w, tb, torch.zeros(4, 6, 1), gx[0], torch.zeros(4, 6, 1), 10, O.replay_noise(list(noise)))
        for gemm in ("f16x3", "f32", "f16x2", "f16"):
            got = c.sample(gx.to(dev()), None, 1, 1, noise=noise.to(dev()), gemm=gemm)[0, :, :, :, 0].cpu()
            # the reduced-precision operand modes are within the budget on the default-init network only
            if gemm in ("f16x3", "f32") or c is ctx:
                assert rel_err(got, want) < RTOL, gemm
            else:
                assert bool(torch.isfinite(got).all())
    # the device counter says which routine ran: default-init weights stay on the fast path, the 2^x'-overflowing ones
    # fall through to the exact routine on (nearly) every step
    steps, safe = ctx.chain_stats()
    assert steps > 0 and safe == 0, (steps, safe)
    steps_h, safe_h = ctx_huge.chain_stats()
    assert safe_h > 0 and safe_h <= steps_h, (steps_h, safe_h)


def test_ragged_and_empty_batches():
    sd, nf, steps = O.random_state_dict(1, 10, 100, 100, 50, seed=5), 1, 10
    ctx = make_ctx(sd, nf, steps)
    out = ctx.sample(torch.ones(0, 3, 4, 1, device=dev()), None, 4, 4, seed=1)
    assert tuple(out.shape) == (0, 3, 4, 1, 4)
    # 1*1*3*5 = 15 chains: less than one 32-chain tile; 3*7*11*3 = 693: ragged last tile
    for shape, S in (((1, 1, 3, 1), 5), ((3, 7, 11, 1), 3)):
        g = torch.Generator().manual_seed(7)
        gx = torch.rand(*shape, generator=g) + 0.1
        W, B, Ol, _ = shape
        noise = torch.randn(W, 1, steps, B * S, Ol, 1, generator=g)
        got = ctx.sample(gx.to(dev()), None, S, S, noise=noise.to(dev())).cpu()
        for w in range(W):
            rows = gx[w].unsqueeze(1).expand(B, S, Ol, 1).reshape(B * S, Ol, 1)
            want = O.sample_loop(sd, O.schedule_tables(steps), torch.zeros_like(rows), rows, torch.zeros_like(rows), steps,
                                 O.replay_noise(list(noise[w, 0])))
            want = want.reshape(B, S, Ol, 1).permute(0, 2, 3, 1)
            assert rel_err(got[w], want) < RTOL
    from mpv_amd._native import MpvError
    with pytest.raises(MpvError):
        ctx.sample(torch.ones(1, 1, 4, 1, device=dev()), None, 6, 4, seed=1)      # S % P != 0


# ---- harness level: the reference's own window loops, reproduced under the reference's seeds ---------
def _reseed_after(monkeypatch, module, name, seed):
    """The reference run seeded torch AFTER the model existed; model construction draws from the same
    generator (nn.Linear init), so re-seed right after the harness has loaded the model."""
    orig = getattr(module, name)

    def wrapped(*a, **k):
        out = orig(*a, **k)
        torch.manual_seed(seed)
        return out

    monkeypatch.setattr(module, name, wrapped)


def test_slbp_sensitivity_ews_matches_reference_curve(tmp_path, golden_cases, ckpt_wofx, monkeypatch):
    """BASELINE configs[0] shape (SLBP single series, 8 windows, N=4) on the real wo_fx checkpoint (T=20)."""
    from helpers import WOFX_NET, write_model_dir
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    root = tmp_path / "sweep"
    cfg = write_model_dir(root / "models" / "tmp", ckpt_wofx, WOFX_NET)
    (root / "models" / "tmp" / "model_trained").rename(root / "models" / "wofx")
    (root / "models" / "tmp" / "model_trained.yaml").rename(root / "models" / "wofx.yaml")
    series, time = T(golden_cases["slbp_series"]), T(golden_cases["slbp_time"])
    _reseed_after(monkeypatch, H, "load_sensitivity_model", 21)
    res = H.slbp_sensitivity_ews(root, "wofx", series, time, sample_window_step=10, max_windows=8, device=dev(),
                                 infer_params={"n_z_samples": 4, "parallel_sample": 4, "noise_source": "torch_cpu"})
    assert len(res["pred_future_list"]) == 8 and tuple(res["pred_future_list"][0].shape) == (200, 2, 4)
    assert rel_err(torch.stack(res["pred_future_list"]), golden_cases["slbp_pred"]) < RTOL
    np.testing.assert_allclose(np.array(res["mpv"], dtype=np.float64), golden_cases["slbp_mpv0"], rtol=RTOL)
    np.testing.assert_allclose(np.array(res["prediction_error"], dtype=np.float64), golden_cases["slbp_err0"], rtol=RTOL, atol=1e-6)
    assert np.array_equal(res["time_points"][:8], golden_cases["slbp_tp"])
    assert H._load_tensor_list(res["cache_path"])[3].shape == (200, 2, 4)              # reference cache format
    mpv1, _ = H.summarize_slbp_sensitivity(res["pred_future_list"], (), pred_dim=1)
    np.testing.assert_allclose(np.array(mpv1, dtype=np.float64), golden_cases["slbp_mpv1"], rtol=RTOL)
    again = H.slbp_sensitivity_ews(root, "wofx", series, time, sample_window_step=10, max_windows=8, device=dev())
    assert all(torch.equal(a, b) for a, b in zip(again["pred_future_list"], res["pred_future_list"]))   # served from cache


def test_uncertainty_ews_network_curve_matches_reference(tmp_path, golden_cases, monkeypatch):
    """uncertainty_ews end to end (scaler, windows, sigma head, sampler, MPV reduce, caches) == reference run."""
    from helpers import WOFX_NET, write_model_dir
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    sd["scaler_mean"] = T(golden_cases["eval_seed_scaler_mean"])
    sd["scaler_std"] = T(golden_cases["eval_seed_scaler_std"])
    net = dict(WOFX_NET, diffusion_steps=50, rolling_length=50, windows=100, pred_len=100, dataset_nf=1)
    mdir = tmp_path / "model"
    write_model_dir(mdir, sd, net)
    series = T(np.ascontiguousarray(golden_cases["net_series"]))[:, :, 0].t().contiguous()        # [ticks, V]
    torch.save({"ys_dynamic": series, "ts_dynamic": torch.arange(series.shape[0]) * 0.1, "tp_values": torch.zeros(1)},
               tmp_path / "biomass_run.pt")
    _reseed_after(monkeypatch, H, "load_model_from_dir", 22)
    res = H.uncertainty_ews(model_save_file=mdir, data_file=tmp_path / "biomass_run.pt", dynamic_type="biomass",
                            sampling_t=0.1, sample_window_step=80, device=dev(), uncertainty_method="both",
                            infer_params={"n_z_samples": 8, "parallel_sample": 4, "noise_source": "torch_cpu"})
    assert len(res["ews"]) == 3 and res["uncertainty_source"] == "sampling"
    assert rel_err(torch.stack(res["pred_future_list"]), golden_cases["net_pred"]) < RTOL
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), golden_cases["net_ews"], rtol=RTOL)
    np.testing.assert_allclose(np.array(res["pred_mean"], dtype=np.float64), golden_cases["net_pred_mean"], rtol=RTOL)
    np.testing.assert_allclose(res["time_points"], golden_cases["net_tp"])
    np.testing.assert_allclose(np.array(res["nsdiff_g"]["ews"], dtype=np.float64), golden_cases["net_g_ews"], rtol=1e-5)
    # cached second call: summaries recomputed on device from the cache file, same numbers
    res2 = H.uncertainty_ews(model_save_file=mdir, data_file=tmp_path / "biomass_run.pt", dynamic_type="biomass",
                             sampling_t=0.1, device=dev(), load_model_when_cached=True)
    assert res2["sample_window_step"] == 54 or len(res2["ews"]) == 3
    np.testing.assert_allclose(np.array(res2["ews"], dtype=np.float64), golden_cases["net_ews"], rtol=RTOL)


def test_philox_default_mode_statistics_and_repeatability(ckpt_wofx):
    """Default (on-device Philox) sampling: repeatable under torch.manual_seed and statistically
    consistent with the injected-noise path (same model, MPV within sampling error)."""
    model = _variant_model(ckpt_wofx, "cond_var", 2, 20, 200, 200, 100, 64, 64)
    g = torch.Generator().manual_seed(4)
    batch = torch.randn(6, 200, 2, generator=g)
    torch.manual_seed(7)
    a, _ = model.evaluation_step(batch)
    torch.manual_seed(7)
    b, _ = model.evaluation_step(batch)
    c, _ = model.evaluation_step(batch)
    assert torch.equal(a, b) and not torch.equal(a, c)
    ref_model = _variant_model(ckpt_wofx, "cond_var", 2, 20, 200, 200, 100, 64, 64, noise_source="torch_cpu")
    torch.manual_seed(8)
    d, _ = ref_model.evaluation_step(batch)
    va, vd = a.var(dim=-1, unbiased=False).mean().item(), d.var(dim=-1, unbiased=False).mean().item()
    assert abs(va - vd) / vd < 0.05, (va, vd)
    assert abs(a.mean().item() - d.mean().item()) < 0.05 * d.std().item()


@pytest.mark.parametrize("select", ["wo_UANS", "cond_mean", "NsDiff_model"])
def test_attached_backbone_variants_match_oracle(ckpt_wofx, select):
    """The f(x)-using selects with a caller-supplied backbone: y_0_hat comes from the attached torch module, the rest
    (sigma head with/without EPS or gx = 1, sampler around y_0_hat) is native and equals the oracle on shared noise."""
    from mpv_amd.models.models import diffusion_models

    class Backbone(torch.nn.Module):                       # stand-in f(x): any module with the reference's signature
        def __init__(self):
            super().__init__()
            self.mix = torch.nn.Linear(200 + 300, 200, bias=True)

        def forward(self, x, dec_inp):
            z = torch.cat([x, dec_inp], dim=1).transpose(1, 2)              # [B, F, 500]
            return torch.tanh(self.mix(z)).transpose(1, 2) * 0.3, None

    torch.manual_seed(3)
    f = Backbone()
    net = dict(device=dev(), scaler_type="StandardScaler", task_model="x", rolling_length=100, diffusion_schedule="linear",
               diffusion_steps=20, beta_start=1e-4, beta_end=0.02, n_z_samples=4, parallel_sample=2, windows=200, pred_len=200,
               dataset_nf=2, load_pretrain=False)
    task = "NsDiff" if select == "NsDiff_model" else "NsDiff_model_variants"
    m = diffusion_models(task, net, train_model_select=select).to(dev())
    own = set(m.state_dict())
    m.load_state_dict({k: v for k, v in ckpt_wofx.items() if k in own})
    m.attach_backbone(f.to(dev())).eval()
    g = torch.Generator().manual_seed(8)
    batch = torch.randn(2, 200, 2, generator=g) * 0.5
    cfg = O.NsDiffConfig(200, 200, 2, 100, 20, n_z_samples=4, parallel_sample=2, variant=select)
    with torch.no_grad():
        dec = torch.cat([batch[:, -100:, :], torch.zeros(2, 200, 2)], dim=1)
        y0 = f.cpu()(batch, dec)[0]
    f.to(dev())
    rec = []
    torch.manual_seed(6)
    want = O.evaluation_step(ckpt_wofx, cfg, batch, y0hat=y0, record=rec)
    outs, _ = m.evaluation_step(batch, noise=torch.cat([z.reshape(-1) for z in rec]))
    err = rel_err(outs, want)
    print("attached backbone [%s] rel err %.3e" % (select, err))
    assert err < RTOL


# ---- BASELINE configs[1] at full size: properties that do not need the oracle ---------------------------------
@pytest.mark.parametrize("V,S,W", [(100, 32, 6), (500, 64, 2)])
def test_full_size_c2_properties(V, S, W):
    """BASELINE configs[1] (100 nodes, N = 32) and configs[3] (500 nodes, N = 64) per-GPU shapes, windows = pred_len = 100,
    T = 50, several windows in one launch:
    finite, bitwise repeatable under the same seed, different under another; the device MPV reduction equals a
    float64 torch reduction of the very samples it consumed; f16x3 and exact-fp32 GEMM modes agree on identical
    Philox noise to the parity tolerance."""
    from mpv_amd import synthetic
    from mpv_amd.models.models import diffusion_models
    Wn, O, Tn = 100, 100, 50
    param = dict(device="cuda:0", dataset_nf=1, windows=Wn, pred_len=O, rolling_length=50, diffusion_steps=Tn,
                 diffusion_schedule="linear", beta_start=1e-4, beta_end=0.02, n_z_samples=S, parallel_sample=S,
                 scaler_type="StandardScaler", task_model="NsDiff_model_variants")
    g = torch.Generator().manual_seed(11)
    series = (torch.cumsum(torch.randn(V, Wn + 5 * (W - 1), 1, generator=g) * 0.05, dim=1) + 3.0)
    proto = diffusion_models("NsDiff_model_variants", dict(param, device="cpu"), train_model_select="cond_var")
    synthetic.seeded_nsdiff_state(proto, series, seed=3)
    sd = {k: v.clone() for k, v in proto.state_dict().items()}
    outs = {}
    for gemm in ("f16x3", "f32"):
        m = diffusion_models("NsDiff_model_variants", dict(param, gemm_mode=gemm), train_model_select="cond_var")
        m.load_state_dict(sd)
        m = m.to("cuda:0").eval()
        samples, gx = m.sample_windows(series, W, first=0, step=5, scale_input=True, seed=77)
        outs[gemm] = samples
        assert tuple(samples.shape) == (W, V, O, 1, S) and bool(torch.isfinite(samples).all())
        if gemm == "f16x3":
            again, _ = m.sample_windows(series, W, first=0, step=5, scale_input=True, seed=77)
            other, _ = m.sample_windows(series, W, first=0, step=5, scale_input=True, seed=78)
            assert torch.equal(samples, again) and not torch.equal(samples, other)
            ctx = m.native_context()
            mpv, pm, _ = ctx.reduce(samples, m.scaler_std, m.scaler_mean)
            x = samples.double() * m.scaler_std.double().view(1, 1, 1, -1, 1) + m.scaler_mean.double().view(1, 1, 1, -1, 1)
            want = x.var(dim=-1, unbiased=False).mean(dim=(1, 2, 3))
            assert rel_err(mpv, want) < 1e-5 and rel_err(pm, x.mean(dim=(1, 2, 3, 4))) < 1e-5
            assert bool((gx > 0).all())
    err = rel_err(outs["f16x3"], outs["f32"])
    print("full-size C2: f16x3 vs exact fp32 on identical Philox noise, max rel err %.3e" % err)
    assert err < RTOL


def test_c_abi_error_paths():
    """Return codes and messages of the C ABI itself (ctypes, no Python-side validation in the way)."""
    import ctypes as C
    from mpv_amd import _native
    lib = _native.load_library()
    assert lib.mpv_nsdiff_sample(None, None, None, None, None, None) < 0                      # NULL ctx
    assert b"ctx is NULL" in (lib.mpv_last_error(None) or b"")
    h = C.c_void_p()
    assert lib.mpv_ctx_create(9999, None, C.byref(h)) < 0 and not h.value                     # bad device ordinal
    assert lib.mpv_ctx_create(0, None, C.byref(h)) == 0 and h.value
    shape = _native.SampleShape(1, 1, 4, 4, 4, 0, 1, 0)
    none_out = _native.SampleOut()
    assert lib.mpv_nsdiff_sample(h, C.byref(shape), None, None, None, C.byref(none_out)) < 0  # nothing loaded
    assert b"mpv_nsdiff_load first" in lib.mpv_last_error(h)
    sd = O.random_state_dict(1, 10, 100, 100, 50, seed=5)
    ctx = make_ctx(sd, 1, 10)
    gx = torch.ones(1, 1, 4, 1, device=dev())
    out = torch.empty(1, 1, 4, 1, 4, device=dev())
    so = _native.SampleOut(samples=out.data_ptr())
    for bad, msg in ((_native.SampleShape(1, 1, 4, 4, 4, 7, 1, 0), b"gemm must be"),
                     (_native.SampleShape(1, 1, 4, 6, 4, 0, 1, 0), b"multiple of chunk"),
                     (_native.SampleShape(-1, 1, 4, 4, 4, 0, 1, 0), b"negative extent")):
        rc = lib.mpv_nsdiff_sample(ctx._h, C.byref(bad), None, C.c_void_p(gx.data_ptr()), None, C.byref(so))
        assert rc < 0 and msg in lib.mpv_last_error(ctx._h), (rc, lib.mpv_last_error(ctx._h))
    ok = _native.SampleShape(1, 1, 4, 4, 4, 0, 1, 0)
    assert lib.mpv_nsdiff_sample(ctx._h, C.byref(ok), None, None, None, C.byref(so)) < 0   # gx NULL, non-empty
    assert lib.mpv_nsdiff_sample(ctx._h, C.byref(ok), None, C.c_void_p(gx.data_ptr()), None, None) < 0            # out NULL
    assert lib.mpv_nsdiff_sample(ctx._h, C.byref(ok), None, C.c_void_p(gx.data_ptr()), None, C.byref(none_out)) < 0   # neither samples nor mpv
    assert b"out->samples and/or out->mpv" in lib.mpv_last_error(ctx._h)
    assert lib.mpv_nsdiff_sample(ctx._h, C.byref(ok), None, C.c_void_p(gx.data_ptr()), None, C.byref(so)) == 0
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    d = _native.NsdiffDesc(0, 1)
    assert lib.mpv_nsdiff_load(ctx._h, C.byref(d), None, None) < 0
    lib.mpv_ctx_destroy(h)


# ---- fused MPV epilogue (ABI v2): summary from the sampler itself, samples optional ---------------------------------------
@pytest.mark.parametrize("nf,S", [(1, 32), (1, 64), (1, 4), (1, 1), (2, 8), (3, 32), (1, 33), (2, 100), (1, 96)])
@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
def test_fused_summary_equals_reduce_of_the_same_samples(nf, S, gemm):
    """mpv_nsdiff_sample with out->mpv: the summary (fused into the split-f16 kernel's epilogue where S allows, sample +
    mpv_reduce otherwise) equals mpv_reduce and a float64 torch reduction of the samples of the SAME call; with
    out->samples == NULL the same Philox seed gives the same curve."""
    steps = 8
    sd = O.random_state_dict(nf, steps, 40, 24, 16, seed=70 + nf)
    ctx = make_ctx(sd, nf, steps)
    g = torch.Generator().manual_seed(S)
    W, B, Ol = 5, 3, 7
    gx = (torch.rand(W, B, Ol, nf, generator=g) + 0.1).to(dev())
    y0 = (torch.randn(W, B, Ol, nf, generator=g) * 0.2).to(dev())
    std = (torch.rand(nf, generator=g) + 0.5).to(dev())
    mean = torch.randn(nf, generator=g).to(dev())
    samples, mpv, pm, feat = ctx.sample(gx, y0, S, S, seed=11, gemm=gemm, summary=True, per_feature=True, std=std, mean=mean)
    mpv_r, pm_r, feat_r = ctx.reduce(samples, std, mean, per_feature=True)
    x = samples.double() * std.double().view(1, 1, 1, -1, 1) + mean.double().view(1, 1, 1, -1, 1)
    var = x.var(dim=-1, unbiased=False)
    assert rel_err(mpv, var.mean(dim=(1, 2, 3))) < 2e-6 and rel_err(feat, var.mean(dim=(1, 2))) < 2e-6
    assert rel_err(pm, x.mean(dim=(1, 2, 3, 4))) < 2e-6
    assert rel_err(mpv, mpv_r) < 2e-6 and rel_err(pm, pm_r) < 2e-6 and rel_err(feat, feat_r) < 2e-6
    none, mpv2, pm2, _ = ctx.sample(gx, y0, S, S, seed=11, gemm=gemm, summary=True, want_samples=False, std=std, mean=mean)
    assert none is None and rel_err(mpv2, mpv) < 1e-6 and rel_err(pm2, pm) < 1e-6
    _, mpv3, _, _ = ctx.sample(gx, y0, S, S, seed=12, gemm=gemm, summary=True, want_samples=False)      # other seed, no scaling
    assert S == 1 or not torch.equal(mpv3, mpv2)


def test_fused_summary_over_window_groups_and_tmdm(golden_tmdm):
    """More windows than the workgroup's LDS accumulator table holds (F = 3: 20 per launch) -> several launches whose
    Philox chains continue where the previous group stopped: identical to the one-launch sample tensor."""
    nf, steps, S = 3, 6, 32
    sd = O.random_state_dict(nf, steps, 40, 24, 16, seed=9)
    ctx = make_ctx(sd, nf, steps)
    g = torch.Generator().manual_seed(1)
    W, B, Ol = 47, 2, 3
    gx = (torch.rand(W, B, Ol, nf, generator=g) + 0.1).to(dev())
    whole = ctx.sample(gx, None, S, S, seed=5)                                    # one launch, no summary
    samples, mpv, pm, _ = ctx.sample(gx, None, S, S, seed=5, summary=True)        # grouped launches
    assert torch.equal(samples, whole)
    want = whole.double().var(dim=-1, unbiased=False).mean(dim=(1, 2, 3))
    assert rel_err(mpv, want) < 2e-6 and rel_err(pm, whole.double().mean(dim=(1, 2, 3, 4))) < 2e-6


def test_one_full_size_c2_window_against_the_oracle():
    """BASELINE configs[1] at FULL size for one window (100 nodes x 100 horizon x 32 samples, T = 50): the default
    split-f16 kernel against the CPU oracle on injected noise, elementwise, and the fused MPV against the oracle's."""
    from mpv_amd import synthetic
    sd0 = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    model = _variant_model(sd0, "cond_var", 1, 50, 100, 100, 50, 32, 32)
    g = torch.Generator().manual_seed(21)
    series = torch.cumsum(torch.randn(100, 100, 1, generator=g) * 0.05, dim=1) + 3.0
    synthetic.seeded_nsdiff_state(model, series, seed=123)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    cfg = O.NsDiffConfig(100, 100, 1, 50, 50, n_z_samples=32, parallel_sample=32)
    x = O.scale(series, sd["scaler_mean"], sd["scaler_std"])
    rec = []
    torch.manual_seed(3)
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    want = O.evaluation_step(sd, cfg, x, record=rec)
    outs, _ = model.evaluation_step(x, noise=torch.cat([z.reshape(-1) for z in rec]))
    err = rel_err(outs, want)
    m_w, v_w = O.mpv_network(want, sd["scaler_mean"], sd["scaler_std"])
    ctx = model.native_context()
    mpv, pm, _ = ctx.reduce(outs.to(dev()).unsqueeze(0).contiguous(), model.scaler_std.float().contiguous(), model.scaler_mean.float().contiguous())
    print("full-size C2 window vs oracle: samples elementwise rel err %.3e, MPV rel err %.3e" % (err, abs(mpv.item() - v_w) / abs(v_w)))
    assert err < RTOL
    assert abs(mpv.item() - v_w) <= 1e-5 * abs(v_w) and abs(pm.item() - m_w) <= 1e-5 * abs(m_w) + 1e-6


@pytest.mark.parametrize("tag", ["wofx", "sim"])
def test_trained_checkpoints_stay_on_the_fast_path(ckpt_wofx, ckpt_sim, tag):
    """Both shipped sampler checkpoints (trained weights, T = 20) on a C2-like batch: how many reverse steps fail the
    end-of-step guard of the split-f16 kernel and are re-run through the exact routine (device counter)."""
    sd, nf, rolling = (ckpt_wofx, 2, 100) if tag == "wofx" else (ckpt_sim, 1, 50)
    model = _variant_model(sd, "cond_var", nf, 20, 200, 200, rolling, 32, 32)
    g = torch.Generator().manual_seed(2)
    series = torch.cumsum(torch.randn(16, 260, nf, generator=g) * 0.1, dim=1)
    ctx = model.native_context()
    ctx.chain_stats(reset=True)
    samples, _ = model.sample_windows(series, 4, first=0, step=20, scale_input=False, seed=3)
    steps, safe = ctx.chain_stats()
    print("trained checkpoint %s: %d of %d tile-steps re-run through the exact routine" % (tag, safe, steps))
    assert bool(torch.isfinite(samples).all()) and steps == (4 * 16 * 200 * 32 // 32) * 20
    assert safe <= steps // 100


@pytest.mark.parametrize("nf,windows,rolling,pred_len", [(2, 1000, 50, 200), (2, 200, 100, 1000), (1, 3000, 100, 100), (4, 1200, 64, 64)])
def test_sigma_head_long_windows_fit_the_lds_budget(nf, windows, rolling, pred_len):
    """Shipped sweep configs (ews_results/NsDiff_windows/models/dataset__w1000p200st100.yaml: F = 2, windows 1000,
    rolling 50; NsDiff_pred ... w200p1000) and longer: rows per workgroup are chosen from the LDS budget (reference
    g_backbone.py:49-72 has no such limit)."""
    sd = O.random_state_dict(nf, 5, windows, pred_len, rolling, seed=windows)
    ctx = make_ctx(sd, nf, 5, sigma=(windows, rolling, pred_len))
    g = torch.Generator().manual_seed(windows)
    x = torch.cumsum(torch.randn(5, windows, nf, generator=g) * 0.1, dim=1)
    got = ctx.sigma_head(x.to(dev()), 1, 0, 1)[0]
    want = O.sigma_head(sd, x, rolling, pred_len)
    err = rel_err(got, want)
    print("sigma head F=%d windows=%d rolling=%d O=%d: rel err %.3e" % (nf, windows, rolling, pred_len, err))
    assert err < 1e-5


def test_uncertainty_ews_curve_only_mode(tmp_path, golden_cases, monkeypatch):
    """store_samples=False: same curve as the reference run, nothing cached, no sample list (fused summary epilogue)."""
    from helpers import WOFX_NET, write_model_dir
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    sd["scaler_mean"] = T(golden_cases["eval_seed_scaler_mean"])
    sd["scaler_std"] = T(golden_cases["eval_seed_scaler_std"])
    net = dict(WOFX_NET, diffusion_steps=50, rolling_length=50, windows=100, pred_len=100, dataset_nf=1)
    mdir = tmp_path / "model"
    write_model_dir(mdir, sd, net)
    series = T(np.ascontiguousarray(golden_cases["net_series"]))[:, :, 0].t().contiguous()
    _reseed_after(monkeypatch, H, "load_model_from_dir", 22)
    res = H.uncertainty_ews(model_save_file=mdir, torch_time_series=series, time_data=torch.arange(series.shape[0]) * 0.1,
                            dynamic_type="biomass", sampling_t=0.1, sample_window_step=80, device=dev(), save_nsdiff_g=False,
                            infer_params={"n_z_samples": 8, "parallel_sample": 4, "noise_source": "torch_cpu"}, store_samples=False)
    assert res["pred_future_list"] is None and not (mdir / "data.pt").exists()
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), golden_cases["net_ews"], rtol=RTOL)
    np.testing.assert_allclose(np.array(res["pred_mean"], dtype=np.float64), golden_cases["net_pred_mean"], rtol=RTOL)
    np.testing.assert_allclose(res["time_points"], golden_cases["net_tp"])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sample_axis_shards_union_is_the_unsplit_run(world):
    """The device side of the harness's sample-axis split (fewer windows than ranks): rank r samples ALL windows with samples
    [r S/world, (r+1) S/world) -- whole parallel_sample chunks (world 2), or a slice of one chunk (world 4, 8) -- from the shared
    reference stream; the ranks' blocks side by side are the single-process samples (the collectives around it: tests/test_distributed_gloo.py)."""
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    sd = O.random_state_dict(2, 6, 24, 12, 8, seed=4)
    sd["scaler_mean"], sd["scaler_std"] = torch.tensor([0.3, -0.2]), torch.tensor([1.7, 0.6])
    model = _variant_model(sd, "cond_var", 2, 6, 24, 12, 8, 8, 4, noise_source="torch_cpu")
    g = torch.Generator().manual_seed(3)
    windows = [torch.randn(5, 24, 2, generator=g).cumsum(1) * 0.2 for _ in range(3)]
    torch.manual_seed(31)
    whole, _, _ = H._native_shard(model, windows, 12, True)                          # [3, 5, 12, 2, 8]
    parts = []
    for r in range(world):
        torch.manual_seed(31)
        parts.append(H._native_sample_shard(model, windows, 12, True, r * (8 // world), (r + 1) * (8 // world), seed=0).cpu())
        assert tuple(parts[-1].shape) == (3, 5, 12, 2, 8 // world)
    assert int(model.configs.n_z_samples) == 8 and int(model.configs.parallel_sample) == 4      # restored
    assert rel_err(torch.cat(parts, dim=-1), whole) < 1e-5
