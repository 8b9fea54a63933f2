"""On-device biomass Euler-Maruyama generator (mpv_biomass_simulate) against its float32 CPU restatement."""
import numpy as np
import pytest
import torch

from oracle import biomass_oracle as BO


def test_csr_matches_dense_adjacency():
    from mpv_amd import synthetic
    adj = synthetic.ba_graph(30, m=3, seed=4)
    rp, ci = synthetic.adjacency_csr(adj)
    rp2, ci2 = BO.csr(adj)
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    assert rp[-1] == int(adj.sum()) and np.array_equal(np.diff(rp), adj.sum(axis=1).astype(np.int32))
    for n in (0, 7, 29):
        assert np.array_equal(ci[rp[n]:rp[n + 1]], np.nonzero(adj[n])[0])


@pytest.mark.gpu
@pytest.mark.parametrize("nodes,ticks,burn,stride,eps_end", [(20, 400, 50, 1, 2.7), (100, 3000, 100, 100, 2.7), (300, 64, 0, 7, 1.0)])
def test_device_trajectory_matches_float32_restatement(nodes, ticks, burn, stride, eps_end):
    """Tick-by-tick agreement with the float32 restatement.  (Explicit Euler is only stable while dt * d * degree < 2;
    the 300-node BA graph has hubs beyond that at dt = 0.1, where any scheme's rounding differences grow to O(1), so
    that case runs at dt = 0.02.)"""
    from mpv_amd import _native, synthetic
    ctx = _native.Context(torch.device("cuda", 0))
    par = dict(r=0.7, k=10.0, d=0.5, eta=0.005, dt=0.02 if nodes > 200 else 0.1, eps_start=0.0, eps_end=eps_end)
    adj = synthetic.ba_graph(nodes, m=3, seed=9)
    rp, ci = synthetic.adjacency_csr(adj)
    x0 = np.random.default_rng(9).uniform(0, 1, nodes).astype(np.float32)
    got = ctx.biomass_simulate(torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(x0).cuda()[None],
                               ticks, burn, stride, seed=77, **par)
    assert tuple(got.shape) == (1, nodes, (ticks + stride - 1) // stride, 1)
    want = BO.simulate(adj, x0, ticks, burn, stride, seed=77, **par)
    err = np.abs(got[0, :, :, 0].cpu().numpy() - want).max() / np.abs(want).max()
    print("biomass sim V=%d ticks=%d: max rel err %.3e" % (nodes, ticks, err))
    assert err < 1e-4
    # two trajectories in one launch: the first is unchanged, the second differs (other Philox chains)
    two = ctx.biomass_simulate(torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(),
                               torch.from_numpy(np.stack([x0, x0])).cuda(), ticks, burn, stride, seed=77, **par)
    assert torch.equal(two[0], got[0]) and not torch.equal(two[1], got[0])


@pytest.mark.gpu
def test_full_size_trajectory_is_usable_by_the_path():
    """BASELINE configs[1] inputs generated on device: 100 nodes, 100 000 ticks, stride 100 -> [100, 1000, 1], finite,
    positive biomass that collapses along the harvesting ramp, accepted by the sigma head."""
    from mpv_amd import _native, synthetic
    ctx = _native.Context(torch.device("cuda", 0))
    series, t = synthetic.biomass_trajectory_device(ctx, 100, 100_000, seed=1000, sampling_stride=100)
    assert tuple(series.shape) == (100, 1000, 1) and t.shape == (1000,) and abs(float(t[1]) - 10.0) < 1e-4
    assert bool(torch.isfinite(series).all())
    early, late = float(series[:, :100].mean()), float(series[:, -50:].mean())
    assert early > 3.0 and late < 0.5 * early
    from mpv_amd._native import MpvError
    with pytest.raises(MpvError):
        ctx.biomass_simulate(torch.zeros(3, dtype=torch.int32).cuda(), torch.zeros(1, dtype=torch.int32).cuda(),
                             torch.zeros(1, 5).cuda(), 10, 0, 1, 0.7, 10.0, 0.5, 0.005, 0.1, 0.0, 1.0, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("model,par", [("SIS", dict(p0=0.0, p1=0.0, p2=0.5, eta=1e-4, eps_start=0.2, eps_end=0.0)),
                                       ("neuronal", dict(p0=2.0, p1=1.5, p2=0.0, eta=0.01, eps_start=0.0, eps_end=0.9))])
def test_sis_and_wilson_cowan_generators_match_float32_restatement(model, par):
    """mpv_network_sde_simulate for the other two dynamics of the paper (BASELINE configs[2] / [3] inputs), tick by tick."""
    from mpv_amd import _native, synthetic
    ctx = _native.Context(torch.device("cuda", 0))
    nodes, ticks, burn, stride = 40, 500, 30, 3
    adj = synthetic.ba_graph(nodes, m=3, seed=5) if model == "SIS" else synthetic.er_graph(nodes, p=0.1, seed=5)
    rp, ci = synthetic.adjacency_csr(adj)
    rng = np.random.default_rng(5)
    x0 = (rng.uniform(0.2, 0.8, nodes) if model == "SIS" else rng.uniform(2.0, 6.0, nodes)).astype(np.float32)
    got = ctx.network_sde_simulate(model, torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(x0).cuda()[None],
                                   ticks, burn, stride, dt=0.1, seed=31, **par)
    want = BO.simulate_model(model, adj, x0, ticks, burn, stride, dt=0.1, seed=31, **par)
    err = np.abs(got[0, :, :, 0].cpu().numpy() - want).max() / np.abs(want).max()
    print("%s sim: max rel err %.3e, final mean %.4f" % (model, err, float(want[:, -1].mean())))
    assert tuple(got.shape) == (1, nodes, (ticks + stride - 1) // stride, 1) and err < 1e-4
    bio = ctx.network_sde_simulate("biomass", torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(x0).cuda()[None],
                                   64, 0, 1, 0.7, 10.0, 0.5, 0.005, 0.1, 0.0, 1.0, 9)
    old = ctx.biomass_simulate(torch.from_numpy(rp).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(x0).cuda()[None], 64, 0, 1,
                               0.7, 10.0, 0.5, 0.005, 0.1, 0.0, 1.0, 9)
    assert torch.equal(bio, old)


def test_slbp_restatement_behaves_like_the_lake_model():
    """CPU: the float32 restatement of the SLBP generator (unpinned: torchsde absent) keeps both populations positive and the bream
    stock rises with the nutrient level, the regime shift the sensitivity data sets are built around."""
    x0 = np.array([0.6, 0.3], dtype=np.float32)
    low = BO.simulate_slbp(x0, 3000, 200, 10, 0.1, 0.0, 0.0, seed=3)
    high = BO.simulate_slbp(x0, 3000, 200, 10, 0.1, 3.0, 3.0, seed=3)
    assert low.shape == (300, 2) and np.isfinite(low).all() and np.isfinite(high).all()
    assert (low > 0).all() and (high > 0).all() and high[-1, 0] > low[-1, 0]
    again = BO.simulate_slbp(x0, 3000, 200, 10, 0.1, 0.0, 0.0, seed=3)
    assert np.array_equal(low, again)


@pytest.mark.gpu
def test_slbp_generator_matches_float32_restatement():
    """mpv_slbp_simulate (the SLBP single-series data sets, BASELINE configs[0] input), tick by tick, several trajectories at once."""
    from mpv_amd import _native, synthetic
    ctx = _native.Context(torch.device("cuda", 0))
    rng = np.random.default_rng(8)
    x0 = rng.uniform(0.0, 1.0, (3, 2)).astype(np.float32)
    ticks, burn, stride = 2001, 77, 7
    got = ctx.slbp_simulate(torch.from_numpy(x0).cuda(), ticks, burn, stride, 0.1, 0.2, 2.5, seed=41, D=1e-3).cpu().numpy()
    assert got.shape == (3, (ticks + stride - 1) // stride, 2)
    for traj in range(3):
        want = BO.simulate_slbp(x0[traj], ticks, burn, stride, 0.1, 0.2, 2.5, seed=41, traj=traj, D=1e-3)
        err = np.abs(got[traj] - want).max() / np.abs(want).max()
        print("SLBP trajectory %d: max rel err %.3e, final (bream, pike) = (%.4f, %.4f)" % (traj, err, want[-1, 0], want[-1, 1]))
        assert err < 1e-4
    series, t, n = synthetic.slbp_series_device(ctx, ticks=20_000, seed=2, sampling_stride=100)
    assert tuple(series.shape) == (200, 2) and len(t) == len(n) == 200 and bool(torch.isfinite(series).all()) and n[0] == 0.0
    with pytest.raises(_native.MpvError, match="unknown SLBP"):
        ctx.slbp_simulate(torch.from_numpy(x0).cuda(), 10, 0, 1, 0.1, 0.0, 1.0, seed=1, bogus=1.0)
