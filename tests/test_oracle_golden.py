"""Pin the CPU oracle against vectors recorded from the reference's own code
(tests/golden/make_golden.py).  Tolerances: tables and index work exact; float paths
1e-6 relative (same torch ops on the same CPU, only op grouping differs)."""
import numpy as np
import pytest
import torch

from oracle import nsdiff_oracle as O

T = torch.from_numpy


def close(a, b, rtol=2e-6, atol=1e-7):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("steps,kind", [(10, "linear"), (20, "linear"), (50, "linear"), (8, "quad"), (12, "cosine")])
def test_schedule_tables_bit_exact(golden_schedule, steps, kind):
    tb = O.schedule_tables(steps, kind)
    for name, val in tb.items():
        want = golden_schedule["%s_%d_%s" % (kind, steps, name)]
        assert np.array_equal(val.numpy(), want), name


def test_trailing_variance(golden_cases):
    x = T(golden_cases["rollvar_x"])
    for k in (1, 7, 40):
        close(O.trailing_variance(x, k), golden_cases["rollvar_k%d" % k], rtol=1e-6, atol=1e-9)
    with pytest.raises(ValueError):
        O.trailing_variance(x, 41)
    with pytest.raises(ValueError):
        O.trailing_variance(x[0], 3)


def test_sigma_head(golden_cases, ckpt_wofx):
    close(O.sigma_head(ckpt_wofx, T(golden_cases["ghead_wofx_x"]), 100, 200), golden_cases["ghead_wofx_y"])
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    close(O.sigma_head(sd, T(golden_cases["ghead_seed_x"]), 50, 100), golden_cases["ghead_seed_y"])


@pytest.mark.parametrize("tag,nf,steps", [("seed", 1, 50), ("wofx", 2, 20)])
def test_eps_network_every_t(golden_cases, ckpt_wofx, tag, nf, steps):
    sd = ckpt_wofx if tag == "wofx" else O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    y, y0, gx = (T(golden_cases["eps_%s_%s" % (tag, n)]) for n in ("y", "y0", "gx"))
    for t in range(steps):
        eps, sig = O.eps_network(sd, y, y0, gx, t)
        close(eps, golden_cases["eps_%s_eps" % tag][t])
        close(sig, golden_cases["eps_%s_sig" % tag][t])


def _model(tag, ckpt_wofx):
    if tag == "wofx20":
        return ckpt_wofx, 2, 20
    if tag == "seed50":
        return O.random_state_dict(1, 50, 100, 100, 50, seed=123), 1, 50
    return O.random_state_dict(2, 10, 200, 200, 100, seed=321), 2, 10


@pytest.mark.parametrize("tag", ["seed50", "wofx20", "seed10"])
def test_single_steps_and_loop(golden_cases, ckpt_wofx, tag):
    sd, nf, steps = _model(tag, ckpt_wofx)
    tb = O.schedule_tables(steps)
    y, y0, gx = (T(golden_cases["step_%s_%s" % (tag, n)]) for n in ("y", "y0", "gx"))
    for t in (steps - 1, steps // 2, 1):
        z = T(golden_cases["step_%s_t%d_z" % (tag, t)])
        close(O.reverse_step(sd, tb, y, y0, gx, y0, t, z), golden_cases["step_%s_t%d_out" % (tag, t)], rtol=1e-5, atol=1e-6)
    close(O.final_step(sd, tb, y, y0, gx, y0), golden_cases["step_%s_final" % tag], rtol=1e-5, atol=1e-6)
    noise = T(golden_cases["loop_%s_noise" % tag])
    assert noise.shape[0] == steps                       # z_T + (T-1) step draws
    got = O.sample_loop(sd, tb, y0, gx, y0, steps, O.replay_noise(list(noise)))
    close(got, golden_cases["loop_%s_out" % tag], rtol=2e-5, atol=2e-6)


def _replay_flat(flat, shape_numel):
    flat = T(flat)
    pos = [0]

    def fn(shape):
        n = int(np.prod(shape))
        z = flat[pos[0]: pos[0] + n].reshape(shape)
        pos[0] += n
        return z

    fn.consumed = lambda: pos[0]
    return fn


def test_evaluation_step_real_ckpt_two_chunks(golden_cases, ckpt_wofx):
    cfg = O.NsDiffConfig(200, 200, 2, 100, 20, n_z_samples=4, parallel_sample=2)
    fn = _replay_flat(golden_cases["eval_wofx_noise"], None)
    out = O.evaluation_step(ckpt_wofx, cfg, T(golden_cases["eval_wofx_batch"]), fn)
    assert fn.consumed() == golden_cases["eval_wofx_noise"].size
    assert tuple(out.shape) == (1, 200, 2, 4)
    close(out, golden_cases["eval_wofx_out"], rtol=5e-5, atol=5e-6)


def test_evaluation_step_seeded_network_shape(golden_cases):
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    cfg = O.NsDiffConfig(100, 100, 1, 50, 50, n_z_samples=8, parallel_sample=4)
    fn = _replay_flat(golden_cases["eval_seed_noise"], None)
    out = O.evaluation_step(sd, cfg, T(golden_cases["eval_seed_batch"]), fn)
    assert fn.consumed() == golden_cases["eval_seed_noise"].size
    close(out, golden_cases["eval_seed_out"], rtol=5e-5, atol=5e-6)


def test_evaluation_step_seed_stream_matches_reference(golden_cases):
    """Same torch.manual_seed => the oracle draws the very numbers the reference drew."""
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    cfg = O.NsDiffConfig(100, 100, 1, 50, 50, n_z_samples=8, parallel_sample=4)
    rec = []
    torch.manual_seed(12)
    out = O.evaluation_step(sd, cfg, T(golden_cases["eval_seed_batch"]), record=rec)
    flat = torch.cat([r.reshape(-1) for r in rec]).numpy()
    assert np.array_equal(flat, golden_cases["eval_seed_noise"])
    close(out, golden_cases["eval_seed_out"], rtol=5e-5, atol=5e-6)


def test_evaluation_step_gaussian_variant(golden_cases):
    sd = O.random_state_dict(2, 10, 200, 200, 100, seed=321)
    cfg = O.NsDiffConfig(200, 200, 2, 100, 10, n_z_samples=4, parallel_sample=4, variant="Guassian")
    fn = _replay_flat(golden_cases["eval_gauss_noise"], None)
    out = O.evaluation_step(sd, cfg, T(golden_cases["eval_gauss_batch"]), fn)
    close(out, golden_cases["eval_gauss_out"], rtol=5e-5, atol=5e-6)


def test_sampling_stride_and_windows_exact(golden_cases):
    keys = [str(k) for k in golden_cases["stride_keys"]]
    for k, v in zip(keys, golden_cases["stride_vals"]):
        arg = None if k == "None" else float(k)
        assert O.sampling_stride(arg) == int(v), k
    for i in range(int(golden_cases["win_cases"])):
        L, w, s, cnt = (int(v) for v in golden_cases["win%d_args" % i])
        assert O.window_count(L, w, s) == cnt
        if L >= w:
            assert np.array_equal(O.window_starts(L, w, s), golden_cases["win%d_first" % i])
            series = torch.arange(2 * L, dtype=torch.float32).reshape(2, L, 1)
            wins, tp = O.sliding_windows(series, np.arange(L) * 0.5, w, s)
            assert len(wins) == cnt and all(tuple(x.shape) == (2, w, 1) for x in wins)
            assert np.array_equal(tp, golden_cases["win%d_time" % i])
        else:
            with pytest.raises(ValueError):
                O.sliding_windows(torch.zeros(2, L, 1), np.arange(L), w, s)


def test_slbp_harness_curve(golden_cases, ckpt_wofx):
    """C1 plumbing: 8 SLBP windows, real checkpoint, N=4, B=1 (run_slbp_sensitivity_cache)."""
    series = T(golden_cases["slbp_series"])
    sub = series[:: O.sampling_stride(10)]
    cfg = O.NsDiffConfig(200, 200, 2, 100, 20, n_z_samples=4, parallel_sample=4)
    torch.manual_seed(21)
    rec = []
    for w in range(8):
        x = O.scale(sub[w * 10: w * 10 + 200], ckpt_wofx["scaler_mean"], ckpt_wofx["scaler_std"]).unsqueeze(0)
        out = O.evaluation_step(ckpt_wofx, cfg, x, record=rec if w == 0 else None).squeeze(0)
        close(out, golden_cases["slbp_pred"][w], rtol=5e-5, atol=5e-6)
        assert abs(O.mpv_slbp(out, 0) - golden_cases["slbp_mpv0"][w]) <= 1e-5 * abs(golden_cases["slbp_mpv0"][w])
        assert abs(O.mpv_slbp(out, 1) - golden_cases["slbp_mpv1"][w]) <= 1e-5 * abs(golden_cases["slbp_mpv1"][w])
    assert np.array_equal(torch.cat([r.reshape(-1) for r in rec]).numpy(), golden_cases["slbp_noise0"])


def test_network_harness_curve(golden_cases):
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    sd["scaler_mean"] = T(golden_cases["eval_seed_scaler_mean"])
    sd["scaler_std"] = T(golden_cases["eval_seed_scaler_std"])
    cfg = O.NsDiffConfig(100, 100, 1, 50, 50, n_z_samples=8, parallel_sample=4)
    series = T(golden_cases["net_series"])
    fn = _replay_flat(golden_cases["net_noise"], None)
    preds, means, mpvs = O.mpv_curve(sd, cfg, series, 80, fn)
    assert len(preds) == 3 and fn.consumed() == golden_cases["net_noise"].size
    for w in range(3):
        close(preds[w], golden_cases["net_pred"][w], rtol=5e-5, atol=5e-6)
    np.testing.assert_allclose(mpvs, golden_cases["net_ews"], rtol=1e-5)
    np.testing.assert_allclose(means, golden_cases["net_pred_mean"], rtol=1e-5)
    # gx-only path (run_nsdiff_g_cache: no +EPS) and its summary
    wins, _ = O.sliding_windows(series, np.arange(series.shape[1]), 100, 80)
    for w, win in enumerate(wins):
        gx = O.sigma_head(sd, O.scale(win, sd["scaler_mean"], sd["scaler_std"]), 50, 100)
        close(gx, golden_cases["net_gx"][w])
        m, e = O.gx_summary(gx, 0)
        assert abs(m - golden_cases["net_g_mean"][w]) <= 1e-5 * abs(m)
        assert abs(e - golden_cases["net_g_ews"][w]) <= 1e-5 * abs(e)
