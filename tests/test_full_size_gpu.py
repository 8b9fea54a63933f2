"""Full-size GPU tests (-m gpu) of BASELINE.json configs[2] (C3 DiffusionTS), configs[3] (C4 NsDiff 500 nodes, N = 64) and configs[4]
(C5 DiffSTG): ONE window of each at the size the metric is quoted on -- 3200 sequences x 200 tokens x T = 100 (C3), 3.2 M chains
x T = 50 (C4), 6400 graph-node rows x 50 inference steps (C5) -- i.e. the launches `bench.py --config c3|c4|c5` times.

The CPU oracle cannot run these sizes in seconds, so each test combines
  * size-independent properties on the full launch: finite, bitwise repeatable under a seed, changes with the seed, independent of how
    the launch is cut (rows per pass, windows batched vs one by one), device MPV == float64 torch reduction of the returned samples;
  * an oracle comparison on a SUBSET OF THE SAME LAUNCH: the full launch runs on injected normal draws, and the rows of a few
    sequences / nodes / one graph copy are replayed through oracle/ with exactly their slice of those draws.  Rows couple through
    nothing but the batch normalisers, which the replay takes from the full batch.
A row-index or arena-offset bug that only shows at B = 3200 / 6400 moves the subset rows and fails the comparison."""
import os
import types

import numpy as np
import pytest
import torch

from test_gpu_parity import dev, rel_err

pytestmark = pytest.mark.gpu


def mpv64(samples, std=None, mean=None):
    """summarize_pred_future_list (reference diffusion_model_uncertainy.py:286-303) in float64: [V, O, F, S] -> MPV, pred_mean."""
    x = samples.double().cpu()
    if std is not None:
        x = x * std.double().cpu().view(1, 1, -1, 1) + mean.double().cpu().view(1, 1, -1, 1)
    return float(x.var(dim=-1, unbiased=False).mean()), float(x.mean())


# ---------------------------------------------------------------------------------------------------------------------------------
# C3: DiffusionTS, 100 nodes x N = 32 in one chunk = 3200 sequences, T = 100 DDIM steps with the Langevin infill
# ---------------------------------------------------------------------------------------------------------------------------------
def make_c3(sampling_steps=None):
    from mpv_amd.bench_dts import C3
    from mpv_amd.models.models import diffusion_models
    cfg = dict(C3)
    if sampling_steps is not None:
        cfg["diffusion_steps"] = sampling_steps
    net = dict(device=dev(), scaler_type="StandardScaler", task_model="DiffusionTS", dataset_nf=cfg["nf"], windows=cfg["windows"], pred_len=cfg["pred_len"],
               d_model=cfg["d_model"], n_heads=cfg["n_heads"], n_layer_enc=cfg["n_layer_enc"], n_layer_dec=cfg["n_layer_dec"], timesteps=cfg["timesteps"],
               diffusion_steps=cfg["diffusion_steps"], n_z_samples=cfg["n_z"], parallel_sample=cfg["parallel_sample"])
    torch.manual_seed(123)
    model = diffusion_models("DiffusionTS", net).to(dev()).eval()
    g = torch.Generator().manual_seed(77)
    t = torch.linspace(0, 6.0, cfg["windows"]).view(1, -1, 1)
    phase = torch.rand(cfg["nodes"], 1, 1, generator=g) * 6.28
    hist = (0.6 * torch.sin(t * (1.0 + 0.3 * torch.rand(cfg["nodes"], 1, 1, generator=g)) + phase)
            + 0.05 * torch.randn(cfg["nodes"], cfg["windows"], 1, generator=g)).clamp(-1, 1)          # [V, windows, 1], scaled range
    return model, cfg, hist.unsqueeze(0).to(dev())


@pytest.fixture(scope="module")
def c3():
    return make_c3()


def test_c3_full_window_properties(c3):
    model, cfg, hist = c3
    ctx = model.native_context()
    a = model.sample_windows(hist, seed=1234)                                   # [1, 100, 100, 1, 32]: 3200 sequences, 228 + 128 passes
    assert tuple(a.shape) == (1, cfg["nodes"], cfg["pred_len"], 1, cfg["n_z"]) and bool(torch.isfinite(a).all())
    assert float(a.abs().max()) <= 1.0 + 1e-6                                   # the last step returns the clamped x0 estimate
    mpv, pm, _ = ctx.reduce(a)
    want_mpv, want_pm = mpv64(a[0])
    print("C3 full window: MPV %.6g (float64 reduction %.6g), pred_mean %.6g" % (float(mpv[0]), want_mpv, float(pm[0])))
    assert abs(float(mpv[0]) - want_mpv) <= 1e-5 * abs(want_mpv) and abs(float(pm[0]) - want_pm) <= 1e-5 * max(abs(want_pm), 1e-3)
    assert want_mpv > 0
    os.environ["MPV_DTS_ROWS"] = "800"                                          # the same launch in four row blocks of 800 sequences
    try:
        b = model.sample_windows(hist, seed=1234)
    finally:
        del os.environ["MPV_DTS_ROWS"]
    assert torch.equal(a, b), "rows_per_pass changed the samples (max diff %.3e)" % float((a - b).abs().max())


def _c3_subset_replay(model, cfg, hist, seed):
    """Full 3200-sequence launch on injected draws; sequences (sample s, node v) for s, v < 4 (rows s*100 + v of the sample-major batch)
    replayed through oracle/diffusionts_oracle.py with their slice of the same draws and the FULL batch's normalisers."""
    from oracle import diffusionts_oracle as O
    V, par, L = cfg["nodes"], cfg["parallel_sample"], cfg["windows"] + cfg["pred_len"]
    plan = model._plan()
    draws = 1 + sum(2 + int(r[6]) for r in plan.tolist() if r[1] >= 0)
    g = torch.Generator().manual_seed(seed)
    noise = torch.randn(1, draws, 1, par * V, L, 1, generator=g)                # T = 100: 327 draws x 3200 x 200 = 837 MB
    out = model.sample_windows(hist, noise=noise)[0].cpu()                      # [V, O, 1, S] after the node-major un-tiling quirk
    assert bool(torch.isfinite(out).all())
    flat = out.permute(0, 3, 1, 2).reshape(V * par, cfg["pred_len"], 1)         # row r of the sample-major batch
    ns, nv = 4, 4
    rows = torch.tensor([s * V + v for s in range(ns) for v in range(nv)])
    it = iter(noise[0, :, 0, rows].unbind(0))                                   # [draws][16, L, 1]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ocfg = O.DTSConfig(cfg["windows"], cfg["pred_len"], cfg["nf"], cfg["d_model"], cfg["n_layer_enc"], cfg["n_layer_dec"], cfg["n_heads"],
                       cfg["timesteps"], cfg["diffusion_steps"], n_z_samples=ns, parallel_sample=ns)
    tb = O.schedule_tables(cfg["timesteps"])
    hx = hist[0, :nv].cpu()                                                     # [nv, windows, 1]
    x = torch.cat([hx, torch.zeros(nv, cfg["pred_len"], 1)], dim=1).repeat(ns, 1, 1)       # sample-major, like the reference
    mask = torch.cat([torch.ones(cfg["windows"], 1, dtype=torch.bool), torch.zeros(cfg["pred_len"], 1, dtype=torch.bool)]).expand(ns * nv, -1, -1)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    want = O.fast_sample_infill(sd, ocfg, tb, x * mask, mask, lambda shape: next(it), batch_of=V * par)[:, -cfg["pred_len"]:, :]
    got = flat[rows]
    g_s, w_s = got.reshape(ns, nv, -1), want.reshape(ns, nv, -1)
    mpv_g, mpv_w = float(g_s.double().var(dim=0, unbiased=False).mean()), float(w_s.double().var(dim=0, unbiased=False).mean())
    per_seq = sorted(rel_err(got[j], want[j]) for j in range(ns * nv))       # elementwise error of every replayed sequence on its own
    return per_seq, mpv_g, mpv_w


def test_c3_full_batch_short_plan_subset_against_oracle():
    """B = 3200 sequences (the full C3 batch: every row index, arena offset and batch normaliser of the real launch) with the 5-step plan of
    the reference-run fixture `c3` in tests/golden/diffusionts_cases.npz.  The infill is sign descent around a top-k frequency selection
    (DESIGN section 4): a sequence either follows the oracle's float32 trajectory (elementwise < 1e-2, the gate of tests/test_dts_gpu.py)
    or, after ONE flipped selection, leaves it completely (elementwise O(1); tools/debug/dts_batch_sweep.py shows the same bifurcation
    at every batch shape from 2 x 2 up, and the CPU oracle does it against itself when only the batch shape changes).  A row-index,
    arena-offset or normaliser error moves EVERY replayed sequence, so the gate is on the bulk: the median sequence within 1e-2 and at
    least 10 of the 16 within 1e-2."""
    model, cfg, hist = make_c3(sampling_steps=5)
    per_seq, mpv_g, mpv_w = _c3_subset_replay(model, cfg, hist, seed=4243)
    ok = sum(e < 1e-2 for e in per_seq)
    print("C3 full batch, 5-step plan, 16 replayed sequences vs oracle: per-sequence rel err min %.2e / median %.2e / max %.2e, %d of 16 within "
          "1e-2; subset MPV %.6g vs %.6g" % (per_seq[0], per_seq[len(per_seq) // 2], per_seq[-1], ok, mpv_g, mpv_w))
    assert per_seq[len(per_seq) // 2] < 1e-2 and ok >= 10


def test_c3_full_window_subset_against_oracle_statistically(c3):
    """The same replay at the full T = 100 plan (228 forward + 128 backward passes).  Over 128 sign-descent iterations around a top-k
    frequency selection float32 trajectories decorrelate at the sample level (the reference's own float64 re-run leaves its float32 run
    the same way after 5 steps), so this is a STATISTICAL check: the dispersion of the replayed 16 sequences must agree
    with the oracle's within 10 % (measured 2 %)."""
    model, cfg, hist = c3
    per_seq, mpv_g, mpv_w = _c3_subset_replay(model, cfg, hist, seed=4242)
    print("C3 full launch, T=100, 16 replayed sequences vs oracle: per-sequence rel err min %.2e / median %.2e / max %.2e (chaotic regime); "
          "subset MPV %.6g vs %.6g (rel %.2e)" % (per_seq[0], per_seq[len(per_seq) // 2], per_seq[-1], mpv_g, mpv_w, abs(mpv_g - mpv_w) / abs(mpv_w)))
    assert abs(mpv_g - mpv_w) <= 0.10 * abs(mpv_w)


def test_c3_real_plan_window_against_the_reference_run():
    """A BASELINE configs[2] window on the REAL sampling plan held by the REFERENCE: tests/golden/diffusionts_c3full.npz is the
    reference's own DiffusionTS_model.evaluation_step (DiffusionTS_model.py:72-109 -> DiffusionTS.py:277-310, 359-407) on the C3 model with
    diffusion_steps = 100 -- 228 forward + 128 backward passes -- for 25 nodes x N = 32 = 800 sequences in one chunk under torch.manual_seed (85 minutes
    of 6 CPU threads in the build container; the batch size enters through the Langevin loss normalisers, so 800 is part of the case).  The device run
    replays the same torch CPU stream (noise_source = 'torch_cpu': 327 draws x 800 x 200).  The sampler is sign descent around a top-k frequency
    selection, chaotic at the sample level (the reference's own float64 re-run leaves its float32 trajectory), so sequences either
    stay on the reference's trajectory or leave it; what must agree is the STATISTIC the path exists for -- MPV and pred_mean over the 80 000 (node,
    horizon) rows -- and the test reports the fraction of sequences still within 1e-2 of the reference's."""
    from conftest import GOLDEN
    from mpv_amd.bench_dts import C3
    from mpv_amd.models.models import diffusion_models
    z = np.load(GOLDEN / "diffusionts_c3full.npz")
    nodes, n_z, steps, seed_w, _seed_h, seed_run = (int(v) for v in z["c3full_cfg"])
    net = dict(device=dev(), scaler_type="StandardScaler", task_model="DiffusionTS", dataset_nf=C3["nf"], windows=C3["windows"], pred_len=C3["pred_len"],
               d_model=C3["d_model"], n_heads=C3["n_heads"], n_layer_enc=C3["n_layer_enc"], n_layer_dec=C3["n_layer_dec"], timesteps=C3["timesteps"],
               diffusion_steps=steps, n_z_samples=n_z, parallel_sample=n_z, noise_source="torch_cpu")
    torch.manual_seed(seed_w)
    model = diffusion_models("DiffusionTS", net).to(dev()).eval()
    check = np.array([float(v.double().abs().sum()) for _, v in sorted(model.state_dict().items()) if v.dtype.is_floating_point])
    np.testing.assert_allclose(check, z["c3full_state_checksum"], rtol=1e-12, err_msg="the seeded weights differ from the ones the reference ran")
    hist = torch.from_numpy(z["c3full_hist"])
    torch.manual_seed(seed_run)
    outs, _ = model.evaluation_step(hist)                                         # [25, 100, 1, 32] on the host
    want = torch.from_numpy(z["c3full_out"])
    assert tuple(outs.shape) == tuple(want.shape) and bool(torch.isfinite(outs).all())
    mpv_g, pm_g = mpv64(outs)
    mpv_w, pm_w = float(z["c3full_mpv"]), float(z["c3full_pred_mean"])
    g_seq, w_seq = outs.permute(0, 3, 1, 2).reshape(nodes * n_z, -1), want.permute(0, 3, 1, 2).reshape(nodes * n_z, -1)
    per_seq = np.array([rel_err(g_seq[i], w_seq[i]) for i in range(g_seq.shape[0])])
    frac = float((per_seq < 1e-2).mean())
    row_var_g, row_var_w = outs.double().var(dim=-1, unbiased=False).flatten(), want.double().var(dim=-1, unbiased=False).flatten()
    corr = float(torch.corrcoef(torch.stack([row_var_g, row_var_w]))[0, 1])
    print("C3 real plan (228 + 128 passes), 800 sequences vs the REFERENCE's run: MPV %.6g vs %.6g (rel %.2e), pred_mean %.5g vs %.5g; %.1f %% of the sequences within 1e-2 "
          "(median per-sequence err %.2e); correlation of the per-row variances %.3f" % (mpv_g, mpv_w, abs(mpv_g - mpv_w) / mpv_w, pm_g, pm_w, 100 * frac,
                                                                                      float(np.median(per_seq)), corr))
    # Measured (round 4): MPV 0.829606 vs the reference's 0.826119 (0.42 %), pred_mean 0.0239 vs 0.0214, 28 % of the 800 sequences still within 1e-2 of
    # the reference's after 128 sign-descent iterations, per-row variances correlated at 0.80.  Tolerance: the sequences that left the trajectory are
    # fresh draws of the same distribution, so the two MPVs are two estimates of one population value -- the reference's own float32-vs-float64 runs
    # of the small fixtures differ by 1.6-7 % in MPV.  2 % bounds a systematic error (a wrong normaliser, row index or plan row moves the
    # MPV by tens of percent) and, with >= 15 % of the sequences required on the trajectory, the case cannot pass on statistics alone.
    assert abs(mpv_g - mpv_w) <= 0.02 * mpv_w and abs(pm_g - pm_w) <= 0.01 and frac >= 0.15 and corr >= 0.6


# ---------------------------------------------------------------------------------------------------------------------------------
# C5: DiffSTG, 200-node small-world graph x 32 copies = 6400 node rows, 50 inference steps
# ---------------------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def c5():
    from mpv_amd.bench_stg import C5, small_world
    from mpv_amd.models.models import diffusion_models
    cfg = dict(C5)
    net = dict(device=dev(), scaler_type="StandardScaler", task_model="DiffSTG", F=1, T_h=cfg["windows"], T_p=cfg["pred_len"], Td_h=cfg["Td_h"],
               d_h=cfg["d_h"], channel_multipliers=cfg["channel_multipliers"], n_blocks=cfg["n_blocks"], diffusion_steps=cfg["diffusion_steps"],
               inference_diffusion_steps=cfg["steps"], inference_schedule="cosine", inference_trick="ddim", diffusion_schedule="linear",
               loss_weight_schedule="constant", mask_ratio=0.0, parallel_sampling=cfg["parallel"], sequential_sampling=cfg["sequential"],
               gnn_name="ResGatedGraphConv", gnn_param={"bias": True, "root_weight": True})
    torch.manual_seed(321)
    model = diffusion_models("DiffSTG", net).to(dev()).eval()
    adj = small_world(cfg["nodes"], seed=3000)
    src, dst = np.nonzero(adj)
    edge = torch.from_numpy(np.stack([src, dst])).long()
    g = torch.Generator().manual_seed(78)
    t = torch.linspace(0, 5.0, cfg["windows"]).view(1, 1, -1, 1)
    hist = 0.8 * torch.sin(t + torch.rand(2, cfg["nodes"], 1, 1, generator=g) * 6.28) + 0.1 * torch.randn(2, cfg["nodes"], cfg["windows"], 1, generator=g)
    return model, cfg, net, edge, hist.to(dev())                                 # two windows [2, V, T_h, 1]


def test_c5_full_window_properties(c5):
    model, cfg, net, edge, hist = c5
    ctx = model.native_context()
    a = model.sample_windows(hist[:1], edge, seed=99)                            # [1, 200, 200, 1, 32]
    assert tuple(a.shape) == (1, cfg["nodes"], cfg["windows"] + cfg["pred_len"], 1, cfg["parallel"]) and bool(torch.isfinite(a).all())
    b = model.sample_windows(hist[:1], edge, seed=99)
    c = model.sample_windows(hist[:1], edge, seed=100)
    assert torch.equal(a, b) and not torch.equal(a, c)
    both = model.sample_windows(hist, edge, seed=99)                             # two windows = 12 800 rows in one call
    w1 = model.sample_windows(hist[1:], edge, seed=99)
    assert bool(torch.isfinite(both).all())
    # window 0 of the batched call is the single-window call (same Philox positions); window 1 uses later positions than a call of its own
    assert torch.equal(both[0], a[0]) and tuple(w1.shape) == tuple(a.shape)
    fut = a[:, :, -cfg["pred_len"]:].contiguous()
    mpv, pm, _ = ctx.reduce(fut)
    want_mpv, want_pm = mpv64(fut[0])
    print("C5 full window: MPV %.6g (float64 reduction %.6g)" % (float(mpv[0]), want_mpv))
    assert abs(float(mpv[0]) - want_mpv) <= 1e-5 * abs(want_mpv) and abs(float(pm[0]) - want_pm) <= 1e-5 * max(abs(want_pm), 1e-3)


def test_c5_full_window_first_copies_against_oracle(c5):
    """The full 6400-row launch on injected draws; graph copies 0 and 31 (disjoint copies of the graph couple through nothing) replayed
    through oracle/diffstg_oracle.py as parallel_sampling = 1 runs on their slice of the draws.  Elementwise 1e-4."""
    from oracle import diffstg_oracle as O
    model, cfg, net, edge, hist = c5
    V, P, T = cfg["nodes"], cfg["parallel"], cfg["windows"] + cfg["pred_len"]
    plan = model._plan()
    draws = 1 + int((plan[:, 1] > 0).sum())
    g = torch.Generator().manual_seed(555)
    noise = torch.randn(1, draws, 1, P * V, T, 1, generator=g)
    out = model.sample_windows(hist[:1], edge, noise=noise)[0].cpu()             # [V, T, 1, P]
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    onet = dict(net, device="cpu", parallel_sampling=1, sequential_sampling=1)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    worst = 0.0
    for p in (0, P - 1):
        it = iter(noise[0, :, 0, p * V:(p + 1) * V].unbind(0))                   # rows (w*P + p)*V + v of every draw
        want = O.evaluation_step(sd, onet, hist[0].cpu(), edge, lambda shape: next(it).reshape(shape))   # [V, T, 1, 1]
        err = rel_err(out[..., p], want[..., 0])
        print("C5 full launch, graph copy %d vs oracle: rel err %.3e" % (p, err))
        worst = max(worst, err)
    assert worst < 1e-4


# ---------------------------------------------------------------------------------------------------------------------------------
# C4: NsDiff, 500 nodes x N = 64 x horizon 100 = 3.2 M chains per window, T = 50
# ---------------------------------------------------------------------------------------------------------------------------------
def test_c4_full_window_fused_summary_and_node_subset_against_oracle():
    from oracle import nsdiff_oracle as O
    from mpv_amd import synthetic
    from mpv_amd.bench_c4 import C4
    from mpv_amd.models.models import diffusion_models
    cfg = dict(C4)
    V, S, T_, Ow = cfg["nodes"], cfg["n_z"], cfg["steps"], cfg["pred_len"]
    net = dict(device=dev(), scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=cfg["rolling"], diffusion_schedule="linear",
               diffusion_steps=T_, beta_start=1e-4, beta_end=0.02, n_z_samples=S, parallel_sample=cfg["parallel_sample"], windows=cfg["windows"],
               pred_len=Ow, dataset_nf=1)
    model = diffusion_models("NsDiff_model_variants", net, train_model_select="cond_var").to(dev()).eval()
    g = torch.Generator().manual_seed(5)
    series = torch.cumsum(torch.randn(V, cfg["windows"] + 5, 1, generator=g) * 0.1, dim=1) + 2.0
    synthetic.seeded_nsdiff_state(model, series, seed=123)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = O.scale(series[:, : cfg["windows"]], sd["scaler_mean"], sd["scaler_std"])
    # (1) Philox: two windows in one launch; fused epilogue (S = 64: two 32-sample pieces per row) == float64 reduction of the samples
    samples, _, mpv, pm = model.sample_windows(series.to(dev()), 2, 0, 5, scale_input=True, seed=31, summary=True, want_samples=True)
    assert tuple(samples.shape) == (2, V, Ow, 1, S) and bool(torch.isfinite(samples).all())
    _, _, mpv_only, _ = model.sample_windows(series.to(dev()), 2, 0, 5, scale_input=True, seed=31, summary=True, want_samples=False)
    for w in range(2):
        want_mpv, want_pm = mpv64(samples[w], model.scaler_std, model.scaler_mean)
        assert abs(float(mpv[w]) - want_mpv) <= 2e-6 * abs(want_mpv) and abs(float(pm[w]) - want_pm) <= 2e-6 * abs(want_pm)
    assert torch.equal(mpv, mpv_only)
    # (2) injected draws for the full 500-node window (640 MB); nodes 0..7 and 492..499 replayed through the oracle on their draws
    gz = torch.Generator().manual_seed(6)
    z6 = torch.randn(1, 1, T_, V * S, Ow, 1, generator=gz)                       # [W, C, T, V*P, O, F], rows node-major (v*P + p)
    full, _, mpv_inj, _ = model.sample_windows(x.to(dev()), 1, 0, 1, noise=z6, summary=True, want_samples=True)
    full = full[0].cpu()                                                         # [V, O, 1, S]
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    ocfg = O.NsDiffConfig(cfg["windows"], Ow, 1, cfg["rolling"], T_, n_z_samples=S, parallel_sample=S)
    worst = 0.0
    for lo in (0, V - 8):
        zs = z6[0, 0, :, lo * S:(lo + 8) * S]                                    # [T, 8*S, O, 1] in draw order: z_T first, then one per step
        it = iter(zs.unbind(0))
        want = O.evaluation_step(sd, ocfg, x[lo:lo + 8], noise=lambda shape: next(it).reshape(shape))
        err = rel_err(full[lo:lo + 8], want)
        print("C4 full window, nodes %d..%d vs oracle: rel err %.3e" % (lo, lo + 7, err))
        worst = max(worst, err)
    assert worst < 1e-4
    want_mpv, _ = mpv64(full, model.scaler_std, model.scaler_mean)
    assert abs(float(mpv_inj[0]) - want_mpv) <= 2e-6 * abs(want_mpv)
