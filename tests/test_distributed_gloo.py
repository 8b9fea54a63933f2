"""world_size-2/3 gloo tests of the window sharding + curve reassembly (the N>1 path of the harness)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_windows, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mpv_amd import distributed as D
        from oracle import nsdiff_oracle as O
        assert D.active() and D.world_size() == world and D.rank() == rank
        lo, hi = D.shard_bounds(n_windows)
        # a per-window computation that depends only on the window index (stands in for the sampler):
        g = torch.Generator().manual_seed(5)
        gx_all = torch.rand(n_windows, 3, 7, 1, generator=g) + 0.1
        local = [O.gx_summary(gx_all[w]) for w in range(lo, hi)]
        a = torch.tensor([v[0] for v in local], dtype=torch.float32)
        b = torch.tensor([v[1] for v in local], dtype=torch.float32)
        full_a, full_b = D.gather_curve(n_windows, lo, hi, a, b)
        want = [O.gx_summary(gx_all[w]) for w in range(n_windows)]
        np.testing.assert_allclose(full_a, np.array([v[0] for v in want], dtype=np.float32), rtol=0, atol=0)
        np.testing.assert_allclose(full_b, np.array([v[1] for v in want], dtype=np.float32), rtol=0, atol=0)
        np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([lo, hi]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_windows", [(2, 181), (2, 1), (3, 10), (2, 0)])
def test_sharded_curve_equals_unsharded(tmp_path, world, n_windows):
    mp.spawn(_worker, args=(world, _free_port(), n_windows, str(tmp_path)), nprocs=world, join=True)
    bounds = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    assert bounds[0][0] == 0 and bounds[-1][1] == n_windows
    for a, b in zip(bounds[:-1], bounds[1:]):
        assert a[1] == b[0]                                                # contiguous, no gaps/overlap
    sizes = [int(b[1] - b[0]) for b in bounds]
    assert max(sizes) - min(sizes) <= 1


def test_block_bounds_cover_everything():
    from mpv_amd import distributed as D
    for n in (0, 1, 7, 181, 1000):
        for world in (1, 2, 3, 8):
            spans = [D.block_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
    assert not D.active() and D.rank() == 0 and D.world_size() == 1 and D.shard_bounds(5) == (0, 5)


# ---- the harness itself under sharding: run_evaluation_cache / run_slbp_sensitivity_cache with the CPU oracle injected ----
def _tiny_model(noise_source):
    from mpv_amd.models.models import diffusion_models
    from oracle import nsdiff_oracle as O
    sd = O.random_state_dict(1, 6, 24, 12, 8, seed=4)
    sd["scaler_mean"], sd["scaler_std"] = torch.tensor([0.3]), torch.tensor([1.7])
    net = dict(device="cpu", scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=8,
               diffusion_schedule="linear", diffusion_steps=6, beta_start=1e-4, beta_end=0.02, n_z_samples=4, parallel_sample=2,
               windows=24, pred_len=12, dataset_nf=1, noise_source=noise_source)
    model = diffusion_models("NsDiff_model_variants", net, train_model_select="cond_var")
    model.load_state_dict(sd, strict=True)
    return model.eval(), sd, O.NsDiffConfig(24, 12, 1, 8, 6, n_z_samples=4, parallel_sample=2)


def _oracle_shard(sd, cfg):
    """The CPU oracle as the per-shard computation (same contract as the harness's device shard)."""
    from oracle import nsdiff_oracle as O

    def compute(model, windows, pred_len, scale, keep_samples=True):
        V = int(windows[0].shape[0])
        outs, mpv, pm = [], [], []
        for w in windows:
            x = torch.as_tensor(w).float()
            if scale:
                x = O.scale(x, sd["scaler_mean"], sd["scaler_std"])
            if model.noise_source == "torch_cpu":                # the reference's draw order from the shared CPU stream
                z = model.host_noise(1, V)[0]                    # [C, T, V*P, O, F]
                noise = O.replay_noise([z[c, k] for c in range(z.shape[0]) for k in range(z.shape[1])])
            else:                                                # any per-window deterministic stream
                g = torch.Generator().manual_seed(int(torch.as_tensor(w).double().abs().sum().item() * 1e6) % (2 ** 31))
                noise = lambda shape, g=g: torch.randn(shape, generator=g)
            o = O.evaluation_step(sd, cfg, x, noise=noise)[:, -pred_len:]
            m, v = O.mpv_network(o, sd["scaler_mean"], sd["scaler_std"])
            outs.append(o)
            mpv.append(v)
            pm.append(m)
        return torch.stack(outs), torch.tensor(mpv, dtype=torch.float32), torch.tensor(pm, dtype=torch.float32)
    return compute


def _series(n_windows):
    g = torch.Generator().manual_seed(77)
    s = torch.cumsum(torch.randn(3, 24 + 5 * (n_windows - 1), 1, generator=g) * 0.1, dim=1) + 1.0
    return s, np.arange(s.shape[1], dtype=np.float32) * 0.1


def _harness_worker(rank, world, port, n_windows, noise_source, out_dir):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
        model, sd, cfg = _tiny_model(noise_source)
        H.SHARD_COMPUTE = _oracle_shard(sd, cfg)
        series, time = _series(n_windows)
        views, tp = H.build_sliding_windows(series, time, 24, 5)
        assert len(views) == n_windows
        cache = os.path.join(out_dir, "w%d_cache.pt" % world)
        torch.manual_seed(5)
        preds = H.run_evaluation_cache(model, views, 12, cache, "cpu", force_recompute=True)
        assert len(preds) == n_windows                                         # one entry per window on EVERY rank
        pm, ews = H.summarize_pred_future_list(preds, model=model)
        assert len(ews) == n_windows
        torch.save({"preds": list(preds), "ews": np.array(ews, dtype=np.float64), "pm": np.array(pm, dtype=np.float64)},
                   os.path.join(out_dir, "w%d_r%d.pt" % (world, rank)))
        # single-series (SLBP) variant through the same sharded path
        slbp = [v[0] for v in views]
        torch.manual_seed(6)
        sl = H.run_slbp_sensitivity_cache(model, slbp, os.path.join(out_dir, "w%d_slbp.pt" % world), "cpu", force_recompute=True)
        assert len(sl) == n_windows and tuple(sl[0].shape) == (12, 1, 4)
        if world > 1:
            dist.barrier()
    finally:
        H.SHARD_COMPUTE = None
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("noise_source", ["philox", "torch_cpu"])
@pytest.mark.parametrize("world,n_windows", [(2, 5), (3, 2)])
def test_sharded_harness_cache_equals_unsharded(tmp_path, world, n_windows, noise_source):
    """run_evaluation_cache under world>1 (gloo): the cache file and pred_future_list hold ALL windows, equal the
    single-process cache, and the curve is the same; noise_source='torch_cpu' replays ONE reference stream across ranks."""
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    _harness_worker(0, 1, 0, n_windows, noise_source, str(tmp_path))
    mp.spawn(_harness_worker, args=(world, _free_port(), n_windows, noise_source, str(tmp_path)), nprocs=world, join=True)
    one = torch.load(tmp_path / "w1_r0.pt", weights_only=False)
    cache_one = H._load_tensor_list(tmp_path / "w1_cache.pt")
    cache_n = H._load_tensor_list(tmp_path / ("w%d_cache.pt" % world))
    assert len(cache_one) == len(cache_n) == n_windows
    assert all(torch.equal(a, b) for a, b in zip(cache_one, cache_n))
    assert H.infer_sample_window_step_from_cache(24 + 5 * (n_windows - 1), 24, len(cache_n), 5) == 5
    for r in range(world):
        got = torch.load(tmp_path / ("w%d_r%d.pt" % (world, r)), weights_only=False)
        assert all(torch.equal(a, b) for a, b in zip(one["preds"], got["preds"]))
        np.testing.assert_array_equal(one["ews"], got["ews"])
        np.testing.assert_array_equal(one["pm"], got["pm"])
    slbp_one = H._load_tensor_list(tmp_path / "w1_slbp.pt")
    slbp_n = H._load_tensor_list(tmp_path / ("w%d_slbp.pt" % world))
    assert len(slbp_n) == n_windows and all(torch.equal(a, b) for a, b in zip(slbp_one, slbp_n))


# ---- fewer windows than ranks: the sample axis is split instead ------------------------------------------------------------------
def _oracle_sample_shard(sd):
    """The CPU oracle as one rank's share of a sample-axis split: all windows, samples [s_lo, s_hi) of each, drawn from the shared
    reference stream exactly as the device path slices it."""
    from oracle import nsdiff_oracle as O

    def compute(model, windows, pred_len, scale, s_lo, s_hi):
        V, per, P = int(windows[0].shape[0]), s_hi - s_lo, int(model.configs.parallel_sample)
        outs = []
        for w in windows:
            x = torch.as_tensor(w).float()
            if scale:
                x = O.scale(x, sd["scaler_mean"], sd["scaler_std"])
            z = model.host_noise(1, V)[0]                                        # [C, T, V*P, O, F], the full draw of this window
            C, T = z.shape[0], z.shape[1]
            z = z.reshape(C, T, V, P, z.shape[-2], z.shape[-1])
            if per % P == 0:
                z, p_loc = z[s_lo // P: s_hi // P], P
            else:
                z, p_loc = z[s_lo // P: s_lo // P + 1, :, :, s_lo % P: s_lo % P + per], per
            z = z.reshape(z.shape[0], T, V * p_loc, z.shape[-2], z.shape[-1])
            cfg = O.NsDiffConfig(24, 12, 1, 8, 6, n_z_samples=per, parallel_sample=p_loc)
            outs.append(O.evaluation_step(sd, cfg, x, noise=O.replay_noise([z[c, k] for c in range(z.shape[0]) for k in range(T)]))[:, -pred_len:])
        return torch.stack(outs)
    return compute


def _sample_split_worker(rank, world, port, n_windows, keep, out_dir):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from mpv_amd import distributed as D
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    try:
        model, sd, cfg = _tiny_model("torch_cpu")
        H.SHARD_COMPUTE, H.SAMPLE_SHARD_COMPUTE = _oracle_shard(sd, cfg), _oracle_sample_shard(sd)
        series, time = _series(n_windows)
        views, _ = H.build_sliding_windows(series, time, 24, 5)
        assert D.sample_split(n_windows, 4, 2) == (world > n_windows)
        torch.manual_seed(5)
        preds = H.run_evaluation_cache(model, views, 12, os.path.join(out_dir, "s%d_cache.pt" % world), "cpu", force_recompute=True, store_samples=keep)
        rec = {"ews": np.asarray(preds.ews, dtype=np.float64), "pm": np.asarray(preds.pred_mean, dtype=np.float64), "preds": list(preds)}
        torch.save(rec, os.path.join(out_dir, "s%d_r%d.pt" % (world, rank)))
        if world > 1:
            dist.barrier()
    finally:
        H.SHARD_COMPUTE = H.SAMPLE_SHARD_COMPUTE = None
        if world > 1:
            dist.destroy_process_group()


@pytest.mark.parametrize("world,n_windows,keep", [(2, 1, True), (4, 2, True), (2, 1, False)])
def test_sample_axis_split_equals_unsplit(tmp_path, world, n_windows, keep):
    """W < world: every rank samples ALL windows with n_z_samples / world samples (whole chunks at world 2, single samples of a
    chunk at world 4); merged moments give the single-process curve and the gathered cache holds the same samples in the same order."""
    _sample_split_worker(0, 1, 0, n_windows, keep, str(tmp_path))
    mp.spawn(_sample_split_worker, args=(world, _free_port(), n_windows, keep, str(tmp_path)), nprocs=world, join=True)
    one = torch.load(tmp_path / "s1_r0.pt", weights_only=False)
    for r in range(world):
        got = torch.load(tmp_path / ("s%d_r%d.pt" % (world, r)), weights_only=False)
        np.testing.assert_allclose(got["ews"], one["ews"], rtol=2e-6)
        np.testing.assert_allclose(got["pm"], one["pm"], rtol=2e-6, atol=1e-7)
        assert len(got["preds"]) == (n_windows if keep else 0)
        for a, b in zip(one["preds"], got["preds"]):
            assert tuple(a.shape) == tuple(b.shape) == (3, 12, 1, 4)
            torch.testing.assert_close(b, a, rtol=1e-5, atol=1e-6)              # same draws; the matmul batch differs
    if keep:
        from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
        cache = H._load_tensor_list(tmp_path / ("s%d_cache.pt" % world))
        assert len(cache) == n_windows and tuple(cache[0].shape) == (3, 12, 1, 4)


# ---- sample-axis split of the other families: dispatch and the whole-chunk rule -----------------------------------------------------------
def _dts_split_worker(rank, world, port, out_dir):
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import GOLDEN
    from mpv_amd import distributed as D
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    from oracle import diffusionts_oracle as O
    try:
        z = np.load(GOLDEN / "diffusionts_cases.npz")
        keys = ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec", "n_heads", "timesteps", "diffusion_steps", "n_z_samples", "parallel_sample")
        par = dict(zip(keys, [int(v) for v in z["cfg_small"]]))
        par.update(n_z_samples=4, parallel_sample=2)
        m = diffusion_models("DiffusionTS", dict(par, device="cpu", scaler_type="StandardScaler", noise_source="torch_cpu"))
        pre = "sd_small_"
        sd = {k[len(pre):]: torch.from_numpy(z[k]) for k in z.files if k.startswith(pre)}
        m.load_state_dict(sd, strict=True)
        m.scaler_mean, m.scaler_std = torch.tensor([0.1, -0.2]), torch.tensor([1.5, 0.7])
        sd = {k: v.detach() for k, v in m.state_dict().items()}
        assert H._sample_axis(m) == (4, 2, True)
        assert D.sample_split(1, 4, 2, whole_chunks_only=True) == (world == 2)
        assert not D.sample_split(1, 4, 4, whole_chunks_only=True) or world == 1      # a DiffusionTS chunk is never cut

        def chunk_run(model, windows, pred_len, scale, c_lo, c_hi):
            """The oracle over the chunks [c_lo, c_hi) of every window, drawing the reference's whole stream (the device path's slicing)."""
            V = int(windows[0].shape[0])
            outs = []
            for w in windows:
                x = torch.as_tensor(w).float()
                if scale:
                    x = (x - sd["scaler_mean"]) / sd["scaler_std"]
                zs = model.host_noise(1, V)[:, :, 0]                              # [C, draws, par*V, L, F]
                it = iter([zs[c, k] for c in range(c_lo, c_hi) for k in range(zs.shape[1])])
                cfg = O.DTSConfig(par["windows"], par["pred_len"], par["dataset_nf"], par["d_model"], par["n_layer_enc"], par["n_layer_dec"], par["n_heads"],
                                  par["timesteps"], par["diffusion_steps"], n_z_samples=2 * (c_hi - c_lo), parallel_sample=2)
                outs.append(O.evaluation_step(sd, cfg, x, lambda shape: next(it))[:, -pred_len:])
            return torch.stack(outs)

        def shard(model, windows, pred_len, scale, keep_samples=True, graph=None):
            o = chunk_run(model, windows, pred_len, scale, 0, 2)
            x = o.double() * sd["scaler_std"].double().view(1, 1, 1, -1, 1) + sd["scaler_mean"].double().view(1, 1, 1, -1, 1)
            return o, x.var(dim=-1, unbiased=False).flatten(1).mean(dim=1).float(), x.mean(dim=-1).flatten(1).mean(dim=1).float()

        H.SHARD_COMPUTE = shard
        H.SAMPLE_SHARD_COMPUTE = lambda model, windows, pred_len, scale, s_lo, s_hi: chunk_run(model, windows, pred_len, scale, s_lo // 2, s_hi // 2)
        g = torch.Generator().manual_seed(3)
        windows = [torch.cumsum(torch.randn(2, par["windows"], 2, generator=g) * 0.1, dim=1)]
        torch.manual_seed(5)
        preds = H.run_evaluation_cache(m, windows, par["pred_len"], os.path.join(out_dir, "d%d_cache.pt" % world), "cpu", force_recompute=True)
        torch.save({"ews": np.asarray(preds.ews, dtype=np.float64), "preds": list(preds)}, os.path.join(out_dir, "d%d_r%d.pt" % (world, rank)))
        if world > 1:
            dist.barrier()
    finally:
        H.SHARD_COMPUTE = H.SAMPLE_SHARD_COMPUTE = None
        if world > 1:
            dist.destroy_process_group()


def test_sample_axis_split_of_a_batched_history_family(tmp_path):
    """DiffusionTS with one window on two ranks: each rank runs one whole chunk of the reference's stream; the merged curve and the gathered
    cache equal the single-process run."""
    _dts_split_worker(0, 1, 0, str(tmp_path))
    mp.spawn(_dts_split_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    one = torch.load(tmp_path / "d1_r0.pt", weights_only=False)
    for r in range(2):
        got = torch.load(tmp_path / ("d2_r%d.pt" % r), weights_only=False)
        np.testing.assert_allclose(got["ews"], one["ews"], rtol=2e-6)
        assert len(got["preds"]) == 1 and tuple(got["preds"][0].shape) == tuple(one["preds"][0].shape)
        torch.testing.assert_close(got["preds"][0], one["preds"][0], rtol=1e-5, atol=1e-6)
