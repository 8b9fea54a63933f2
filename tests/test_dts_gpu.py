"""GPU parity tests (-m gpu) of the DiffusionTS HIP path (BASELINE configs[2]) through the C ABI:
transformer forward per block, input gradient (the Langevin infill's autograd), and evaluation_step with the reference's
recorded noise, against fixtures recorded from the reference's own modules (tests/golden/make_golden_diffusionts.py).
Tolerance: north_star's 1e-4 relative on fp32, elementwise (see test_gpu_parity.rel_err)."""
import numpy as np
import pytest
import torch

from test_gpu_parity import rel_err, dev

pytestmark = pytest.mark.gpu
T = torch.from_numpy
RTOL = 1e-4
GOLD = None


def gold():
    global GOLD
    if GOLD is None:
        from conftest import GOLDEN
        GOLD = np.load(GOLDEN / "diffusionts_cases.npz")
    return GOLD


C3 = dict(windows=100, pred_len=100, dataset_nf=1, d_model=64, n_layer_enc=3, n_layer_dec=6, n_heads=4, timesteps=100, diffusion_steps=5,
          n_z_samples=2, parallel_sample=2)
KEYS = ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec", "n_heads", "timesteps", "diffusion_steps",
        "n_z_samples", "parallel_sample")


def build(tag, **over):
    from mpv_amd.models.models import diffusion_models
    z = gold()
    if tag == "c3":
        par = dict(C3)
        torch.manual_seed(11)                                  # the fixture's weights: seeded default init of these very holders
        m = diffusion_models("DiffusionTS", dict(par, device="cpu", scaler_type="StandardScaler", **over))
    else:
        par = dict(zip(KEYS, [int(v) for v in z["cfg_%s" % tag]]))
        m = diffusion_models("DiffusionTS", dict(par, device="cpu", scaler_type="StandardScaler", **over))
        pre = "sd_%s_" % tag
        m.load_state_dict({k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}, strict=True)
    return m.to(dev()).eval(), par


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", ["small", "wide", "c3"])
def test_transformer_forward_matches_reference(tag, gemm):
    m, par = build(tag, gemm_mode=gemm)
    z = gold()
    ctx = m.native_context()
    x = T(z["x_%s" % tag]).to(dev())
    ts = (3, 50, 97) if tag == "c3" else (0, par["timesteps"] // 2, par["timesteps"] - 1)
    for t in ts:
        trend, season = ctx.dts_forward(x, t)
        e1, e2 = rel_err(trend, z["tf_%s_t%d_trend" % (tag, t)]), rel_err(season, z["tf_%s_t%d_season" % (tag, t)])
        print("DiffusionTS forward %s t=%d: trend rel err %.3e, season rel err %.3e" % (tag, t, e1, e2))
        assert e1 < RTOL and e2 < RTOL


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", ["small", "wide", "c3"])
def test_blocks_and_input_gradient_match_reference(tag, gemm):
    """Per-block outputs (encoder blocks, decoder blocks with their mean / trend / season residuals) and the vector-Jacobian
    product of model_output w.r.t. the input, plus the gradient reaching the encoder output.  gemm: the attention products as split-f16 on
    the f16 matrix pipe (default; head width 16 only: 'c3' and 'wide' -- 'small' has 8-wide heads and runs the fp32 kernels either way) or fp32."""
    m, par = build(tag, gemm_mode=gemm)
    z = gold()
    ctx = m.native_context()
    x = T(z["x_%s" % tag]).to(dev())
    t = int(z["blk_%s_t" % tag])
    B, L, D, F = x.shape[0], x.shape[1], par["d_model"], par["dataset_nf"]
    g_out = T(z["vjp_%s_gout" % tag]).to(dev())
    gx = ctx.dts_vjp(x, t, g_out)
    worst = 0.0
    for i in range(par["n_layer_enc"]):
        worst = max(worst, rel_err(ctx.dts_debug("enc_x", i + 1, (B, L, D)), z["blk_%s_enc%d" % (tag, i)]))
    assert worst < RTOL, "encoder blocks %.3e" % worst
    trend_sum, season_sum = 0, 0
    for i in range(par["n_layer_dec"]):
        e = rel_err(ctx.dts_debug("dec_x", i + 1, (B, L, D)), z["blk_%s_dec%d_x" % (tag, i)])
        print("  decoder block %d output rel err %.3e" % (i, e))
        assert e < RTOL, "decoder block %d output %.3e" % (i, e)
        trend_sum = trend_sum + z["blk_%s_dec%d_trend" % (tag, i)]
        season_sum = season_sum + z["blk_%s_dec%d_season" % (tag, i)]
    means = np.concatenate([z["blk_%s_dec%d_mean" % (tag, i)] for i in range(par["n_layer_dec"])], axis=1)
    e_t = rel_err(ctx.dts_debug("trend_sum", 0, (B, L, F)), trend_sum)
    e_s = rel_err(ctx.dts_debug("season_sum", 0, (B, L, D)), season_sum)
    e_m = rel_err(ctx.dts_debug("means", 0, (B, par["n_layer_dec"], F)), means)
    print("DiffusionTS %s blocks: enc %.3e, trend %.3e, season %.3e, means %.3e" % (tag, worst, e_t, e_s, e_m))
    # season_sum is an un-normalised sum of 6 x top-k DFT amplitudes that combine_s then contracts over 64 channels; as an
    # intermediate it is held to the rounding noise of the model OUTPUT (reference fp32 vs its own fp64 run), not tighter
    out_noise = rel_err(z["vjp_%s_y" % tag], z["vjp_%s_y64" % tag])
    assert e_t < RTOL and e_m < RTOL and e_s < max(RTOL, out_noise)
    trend, season = ctx.dts_forward(x, t)
    assert rel_err(trend + season, z["vjp_%s_y" % tag]) < RTOL
    e_g, e_enc = rel_err(gx, z["vjp_%s_gx" % tag]), rel_err(ctx.dts_debug("g_enc", 0, (B, L, D)), z["vjp_%s_genc" % tag])
    # yardstick: the reference's own modules run in float64 on the same inputs.  Its float32 run is only within `noise` of that
    # (3.0e-4 on the C3-dimension model: LayerNorm / mean removal amplify rounding through 9 blocks, forward and backward), so a
    # float32 implementation is asked to be within 1e-4 of the reference's float32 numbers OR as close to the float64 result as
    # the reference's float32 run is (factor 2: two independent rounding noises).
    noise = rel_err(z["vjp_%s_gx" % tag], z["vjp_%s_gx64" % tag])
    e_g64 = rel_err(gx, z["vjp_%s_gx64" % tag])
    print("DiffusionTS %s input gradient: vs reference fp32 %.3e (encoder-output gradient %.3e), vs reference fp64 %.3e; the "
          "reference's fp32 vs its fp64 %.3e" % (tag, e_g, e_enc, e_g64, noise))
    assert e_g < RTOL or e_g64 <= 2.0 * noise
    assert e_enc < max(RTOL, 2.0 * noise)


@pytest.mark.parametrize("tag", ["wide", "c3"])
def test_one_pass_attention_backward_matches_reference_and_the_two_pass_pair(tag, monkeypatch):
    """MPV_DTS_ATTN_ONE_PASS=1: dQ, dK and dV of a (sequence, head) in one kernel (dts_attn_bwd_h_kernel; round 4 -- not the default, the dQ +
    dK-dV pair is 2 % faster at configs[2]).  Same gradient fixture and gate as the default path, the two paths within 1e-5 of each other (they
    differ only in the order the key tiles' dQ contributions are added), bitwise repeatable, and independent of the batch a sequence sits in."""
    m, par = build(tag)
    z = gold()
    ctx = m.native_context()
    x, t = T(z["x_%s" % tag]).to(dev()), int(z["blk_%s_t" % tag])
    g_out = T(z["vjp_%s_gout" % tag]).to(dev())
    two = ctx.dts_vjp(x, t, g_out).clone()
    monkeypatch.setenv("MPV_DTS_ATTN_ONE_PASS", "1")
    one = ctx.dts_vjp(x, t, g_out).clone()
    again = ctx.dts_vjp(x, t, g_out).clone()
    noise = rel_err(z["vjp_%s_gx" % tag], z["vjp_%s_gx64" % tag])
    e_g, e_g64, e_pair = rel_err(one, z["vjp_%s_gx" % tag]), rel_err(one, z["vjp_%s_gx64" % tag]), rel_err(one, two)
    print("DiffusionTS %s one-pass attention backward: vs reference fp32 %.3e, vs reference fp64 %.3e (its fp32 vs fp64 %.3e), vs the two-pass pair %.3e"
          % (tag, e_g, e_g64, noise, e_pair))
    assert e_g < RTOL or e_g64 <= 2.0 * noise
    assert e_pair < 1e-5 and torch.equal(one, again)
    # the same sequences inside a larger batch (other workgroup indices, another tile rotation): the per-sequence arithmetic must not move.
    # dts_vjp's loss normaliser is the caller's g_out, so rows are independent here.
    reps = 3
    xb, gb = x.repeat(reps, 1, 1), g_out.repeat(reps, 1, 1)
    big = ctx.dts_vjp(xb, t, gb)
    B = x.shape[0]
    for k in range(reps):
        assert torch.equal(big[k * B:(k + 1) * B], one), "copy %d of the batch differs" % k


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", ["small", "wide", "c3"])
def test_evaluation_step_matches_reference(tag, gemm):
    """DiffusionTS_model.evaluation_step (DDIM + Langevin infill, sample-major tiling / node-major un-tiling) with every
    normal draw of the reference run injected in its order."""
    m, par = build(tag, gemm_mode=gemm)
    z = gold()
    batch = T(z["batch_%s" % tag])
    outs, by = m.evaluation_step(batch, noise=T(z["eval_%s_noise" % tag]))
    assert by is None and outs.device.type == "cpu"
    want = z["eval_%s_out" % tag]
    assert tuple(outs.shape) == tuple(want.shape)
    err = rel_err(outs, want)
    mpv_g = outs.var(dim=-1, unbiased=False).mean().item()
    mpv_w = T(want).var(dim=-1, unbiased=False).mean().item()
    # The sampler is sign descent (g / (|g| + 1e-10)) around a top-k frequency selection: the reference's own float64 re-run of
    # the same case on the same draws leaves its float32 trajectory completely on the two larger models (elementwise 2.0 and
    # 1.06, MPV by 7 % and 1.6 %), so sample-level agreement with the float32 run is a property of following ITS rounding, not of
    # accuracy.  Required: the MPV within north_star's 1e-4 of the reference's float32 run; the samples within 1e-2 elementwise.
    want64 = z["eval_%s_out64" % tag]
    mpv_64 = float(want64.var(axis=-1).mean())
    print("DiffusionTS evaluation_step %s: samples vs reference fp32 %.3e; MPV %.6g vs %.6g (reference fp64 run: %.6g, its fp32 "
          "samples vs fp64 %.3e)" % (tag, err, mpv_g, mpv_w, mpv_64, rel_err(want, want64)))
    assert err < 1e-2 and abs(mpv_g - mpv_w) <= RTOL * abs(mpv_w)


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
def test_evaluation_step_on_the_full_100_step_plan_small_model(gemm):
    """The reference's evaluation_step of a small model on the REAL plan (timesteps = diffusion_steps = 100: K = 3 / 2 / 1 / 0 Langevin iterations by t,
    654 recorded draws; tests/golden/make_golden_diffusionts_c3full.py, case t100) with its draws injected."""
    from conftest import GOLDEN
    from mpv_amd.models.models import diffusion_models
    z = np.load(GOLDEN / "diffusionts_c3full.npz")
    par = dict(zip(KEYS, [int(v) for v in z["cfg_t100"]]))
    m = diffusion_models("DiffusionTS", dict(par, device="cpu", scaler_type="StandardScaler", gemm_mode=gemm))
    m.load_state_dict({k[len("sd_t100_"):]: T(z[k]) for k in z.files if k.startswith("sd_t100_")}, strict=True)
    m = m.to(dev()).eval()
    plan = m._plan()
    assert plan.shape[0] == 100 and int(plan[:, 6].sum()) == 128 and sorted(set(int(k) for k in plan[:, 6].tolist())) == [0, 1, 2, 3]
    outs, _ = m.evaluation_step(T(z["batch_t100"]), noise=T(z["eval_t100_noise"]))
    want = z["eval_t100_out"]
    err = rel_err(outs, want)
    mpv_g, mpv_w = outs.var(dim=-1, unbiased=False).mean().item(), T(want).var(dim=-1, unbiased=False).mean().item()
    print("DiffusionTS evaluation_step, 100 / 100-step plan, small model (%s): samples vs reference %.3e; MPV %.6g vs %.6g" % (gemm, err, mpv_g, mpv_w))
    assert err < 1e-2 and abs(mpv_g - mpv_w) <= RTOL * abs(mpv_w)


def test_evaluation_step_torch_cpu_stream_and_philox_mode():
    """noise_source='torch_cpu' under the reference's seed == the reference run; the default Philox mode is repeatable under
    torch.manual_seed, changes with the seed, and several windows in one call equal the windows one by one."""
    z = gold()
    m, par = build("small", noise_source="torch_cpu")
    torch.manual_seed(77)
    outs, _ = m.evaluation_step(T(z["batch_small"]))
    assert rel_err(outs, z["eval_small_out"]) < RTOL
    m, par = build("small")
    torch.manual_seed(5)
    a, _ = m.evaluation_step(T(z["batch_small"]))
    torch.manual_seed(5)
    b, _ = m.evaluation_step(T(z["batch_small"]))
    c, _ = m.evaluation_step(T(z["batch_small"]))
    assert torch.equal(a, b) and not torch.equal(a, c) and bool(torch.isfinite(a).all())
    import os
    os.environ["MPV_DTS_ROWS"] = "2"                           # the batch in row blocks of 2 sequences: same numbers (Philox by position)
    try:
        torch.manual_seed(5)
        d, _ = m.evaluation_step(T(z["batch_small"]))
    finally:
        del os.environ["MPV_DTS_ROWS"]
    assert torch.equal(a, d)
    hist = torch.stack([T(z["batch_small"]), T(z["batch_small"]) * 0.5 + 0.1]).to(dev())           # [W=2, V, windows, F]
    m2, _ = build("small", noise_source="torch_cpu")
    torch.manual_seed(9)
    both = m2.sample_windows(hist)
    torch.manual_seed(9)
    one = [m2.sample_windows(hist[w:w + 1]) for w in range(2)]
    assert rel_err(both[0], one[0][0]) < 1e-6 and rel_err(both[1], one[1][0]) < 1e-6


def test_dts_c_abi_error_paths():
    import ctypes as C
    from mpv_amd import _native
    lib = _native.load_library()
    ctx = _native.Context(dev())
    x = torch.zeros(1, 20, 2, device=dev())
    assert lib.mpv_dts_forward(ctx._h, C.c_void_p(x.data_ptr()), 1, 0, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr())) < 0
    assert b"mpv_dts_load first" in lib.mpv_last_error(ctx._h)
    m, par = build("small")
    c = m.native_context()
    with pytest.raises(_native.MpvError):
        c.dts_forward(torch.zeros(1, 21, 2, device=dev()), 0)
    assert lib.mpv_dts_forward(c._h, C.c_void_p(x.data_ptr()), 1, 999, C.c_void_p(x.data_ptr()), C.c_void_p(x.data_ptr())) < 0
    d = _native.DtsDesc(20, 2, 48, 2, 1, 1, 4, 10, 1)
    assert lib.mpv_dts_load(c._h, C.byref(d), None, 0, None, None, None) < 0


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag,eta", [("eta05", 0.5), ("eta1", 1.0)])
def test_evaluation_step_with_eta_matches_reference(tag, eta, gemm):
    """eta != 0: the DDIM step adds sigma * noise and both Langevin loss terms are divided by sigma (reference DiffusionTS.py:294-300,
    392-395); the reference's evaluation_step on the `small` model with every normal draw injected in its order."""
    from conftest import GOLDEN
    ze = np.load(GOLDEN / "diffusionts_eta_cases.npz")
    m, par = build("small", eta=eta, gemm_mode=gemm)
    outs, _ = m.evaluation_step(T(ze["batch"]), noise=T(ze["eval_%s_noise" % tag]))
    want = ze["eval_%s_out" % tag]
    err = rel_err(outs, want)
    mpv_g, mpv_w = outs.var(dim=-1, unbiased=False).mean().item(), T(want).var(dim=-1, unbiased=False).mean().item()
    print("DiffusionTS evaluation_step eta=%.1f (%s): samples rel err %.3e, MPV %.6g vs %.6g" % (eta, gemm, err, mpv_g, mpv_w))
    assert err < 1e-2 and abs(mpv_g - mpv_w) <= RTOL * abs(mpv_w)
    m0, _ = build("small", eta=eta, noise_source="torch_cpu")
    torch.manual_seed(77)
    o2, _ = m0.evaluation_step(T(ze["batch"]))
    assert rel_err(o2, want) < 1e-2                                   # the reference's own seed replays its run (the step's noise is now consumed)
