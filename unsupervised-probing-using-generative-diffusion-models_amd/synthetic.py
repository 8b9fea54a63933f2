"""Seeded synthetic inputs of the reference's shapes (no datasets travel).

* `ba_graph`            : Barabasi-Albert preferential attachment (the reference draws its test
                          topologies with igraph/networkx; dataset/graph_generate.py:61-84)
* `biomass_trajectory`  : Euler-Maruyama restatement of the harvested-biomass network SDE
                          dx_i = [r x_i (1 - x_i/k) - eps(t) x_i^2/(x_i^2+1) + d sum_j A_ij (x_j - x_i)] dt + eta dW_i
                          (dataset/spdata_sde_biomass_dynamic_gene.py:53-68; r=0.7, k=10, d=0.5,
                          eta=0.005 as in paper_figures/common.py:41), dt = 0.1, burn-in, then a
                          linear ramp of the harvesting pressure eps(t) through the fold.
* `slbp_like_series`    : a seeded two-feature single series for the SLBP plumbing shape.

Record format matches the harness input: {'ys_dynamic': [ticks, V], 'ts_dynamic': [ticks]}.
"""
from __future__ import annotations

import numpy as np
import torch


def ba_graph(n: int, m: int = 3, seed: int = 0) -> np.ndarray:
    """Dense symmetric 0/1 adjacency of a BA graph (n nodes, m edges per new node)."""
    rng = np.random.default_rng(seed)
    adj = np.zeros((n, n), dtype=np.float64)
    targets = list(range(m))
    repeated = []
    for src in range(m, n):
        for t in set(targets):
            adj[src, t] = adj[t, src] = 1.0
        repeated.extend(set(targets))
        repeated.extend([src] * m)
        targets = []
        while len(set(targets)) < m:
            targets.append(repeated[rng.integers(len(repeated))])
    return adj


def biomass_trajectory(nodes: int = 100, ticks: int = 100_000, seed: int = 0, r: float = 0.7, k: float = 10.0,
                       d: float = 0.5, eta: float = 0.005, dt: float = 0.1, burn_ticks: int = 1000,
                       eps_start: float = 0.0, eps_end: float = 2.7, m: int = 3):
    """-> record dict with ys_dynamic [ticks, nodes] float32 and ts_dynamic [ticks]."""
    rng = np.random.default_rng(seed)
    A = ba_graph(nodes, m=m, seed=seed)
    deg = A.sum(axis=1)
    x = rng.uniform(0.0, 1.0, nodes)
    sq = np.sqrt(dt)

    def drift(x, eps):
        x = np.maximum(x, 0.0)
        return r * x * (1 - x / k) - eps * x * x / (x * x + 1) + d * (A @ x - deg * x)

    for _ in range(burn_ticks):
        x = x + drift(x, eps_start) * dt + eta * sq * rng.standard_normal(nodes)
    ys = np.empty((ticks, nodes), dtype=np.float32)
    ramp = np.linspace(eps_start, eps_end, ticks)
    noise = rng.standard_normal((ticks, nodes)) * (eta * sq)
    for i in range(ticks):
        x = x + drift(x, ramp[i]) * dt + noise[i]
        ys[i] = x
    return {"ys_dynamic": torch.from_numpy(ys), "ts_dynamic": torch.arange(ticks, dtype=torch.float32) * dt,
            "tp_values": torch.from_numpy(ramp.astype(np.float32))}


def adjacency_csr(adj: np.ndarray):
    """Dense 0/1 adjacency -> (row_ptr [V+1], col_idx [nnz]) int32, neighbours in increasing order."""
    rows, cols = np.nonzero(adj)
    row_ptr = np.zeros(adj.shape[0] + 1, dtype=np.int64)
    np.add.at(row_ptr, rows + 1, 1)
    return np.cumsum(row_ptr).astype(np.int32), cols.astype(np.int32)


def biomass_trajectory_device(ctx, nodes: int = 100, ticks: int = 100_000, seed: int = 0, sampling_stride: int = 1,
                              r: float = 0.7, k: float = 10.0, d: float = 0.5, eta: float = 0.005, dt: float = 0.1,
                              burn_ticks: int = 1000, eps_start: float = 0.0, eps_end: float = 2.7, m: int = 3):
    """Same model and parameters as `biomass_trajectory`, integrated by the HIP kernel csrc/biomass_sim.hip with the
    sampler's Philox stream; only every `sampling_stride`-th tick is stored, directly as the harness's resident
    series.  -> (series [V, L, 1] on ctx.device, sampled time stamps [L] numpy)."""
    adj = ba_graph(nodes, m=m, seed=seed)
    row_ptr, col_idx = adjacency_csr(adj)
    x0 = np.random.default_rng(seed).uniform(0.0, 1.0, nodes).astype(np.float32)
    series = ctx.biomass_simulate(torch.from_numpy(row_ptr).to(ctx.device), torch.from_numpy(col_idx).to(ctx.device),
                                  torch.from_numpy(x0).to(ctx.device).unsqueeze(0), ticks, burn_ticks, sampling_stride,
                                  r, k, d, eta, dt, eps_start, eps_end, seed)
    return series[0], (np.arange(ticks, dtype=np.float32) * np.float32(dt))[::sampling_stride]


def slbp_like_series(ticks: int = 50_000, seed: int = 0, dt: float = 0.1):
    rng = np.random.default_rng(seed)
    t = np.arange(ticks) * dt
    a = 2.0 + 0.4 * np.sin(t / 40.0) + np.cumsum(rng.normal(0, 0.01, ticks))
    b = 1.0 + 0.2 * np.cos(t / 25.0) + np.cumsum(rng.normal(0, 0.015, ticks)) + 0.0004 * t
    return {"ys_dynamic": torch.tensor(np.stack([a, b], axis=1), dtype=torch.float32),
            "ts_dynamic": torch.tensor(t, dtype=torch.float32), "N_values": torch.zeros(ticks)}


def seeded_nsdiff_state(model, series: torch.Tensor, seed: int = 123):
    """Re-initialise `model` in place with torch.manual_seed(seed) default inits (nn.Linear
    kaiming-uniform, embed ~ U(0,1), LayerNorm 1/0) and fit its scaler on `series` [V, L, F]:
    the shipped checkpoints pin T=20, so other step counts need random weights."""
    gen_state = torch.get_rng_state()
    torch.manual_seed(seed)
    for mod in model.modules():
        if isinstance(mod, (torch.nn.Linear, torch.nn.LayerNorm)):
            mod.reset_parameters()
        elif isinstance(mod, torch.nn.Embedding):
            mod.weight.data.uniform_()
    torch.set_rng_state(gen_state)
    flat = series.reshape(-1, series.shape[-1]).float()
    std = flat.std(dim=0)
    std[std == 0] = 1
    model.scaler_mean = flat.mean(dim=0).to(model.scaler_mean.device)
    model.scaler_std = std.to(model.scaler_std.device)
    model.refresh_native()
    return model


def er_graph(n: int, p: float = 0.05, seed: int = 0) -> np.ndarray:
    """Dense symmetric 0/1 adjacency of an Erdos-Renyi graph (isolated nodes get one random neighbour)."""
    rng = np.random.default_rng(seed)
    upper = np.triu(rng.random((n, n)) < p, 1)
    adj = (upper | upper.T).astype(np.float64)
    for i in np.nonzero(adj.sum(axis=1) == 0)[0]:
        j = int(rng.integers(n - 1))
        j += j >= i
        adj[i, j] = adj[j, i] = 1.0
    return adj


def _euler_maruyama(drift, x0, ticks, burn_ticks, dt, eta, eps_start, eps_end, rng):
    sq = np.sqrt(dt)
    x = x0.copy()
    for _ in range(burn_ticks):
        x = x + drift(x, eps_start) * dt + eta * sq * rng.standard_normal(x.shape[0])
    ys = np.empty((ticks, x.shape[0]), dtype=np.float32)
    ramp = np.linspace(eps_start, eps_end, ticks)
    for i in range(ticks):
        x = x + drift(x, ramp[i]) * dt + eta * sq * rng.standard_normal(x.shape[0])
        ys[i] = x
    return ys, ramp


def sis_trajectory(nodes: int = 100, ticks: int = 10_000, seed: int = 0, d: float = 0.5, eta: float = 1e-4, dt: float = 0.1,
                   burn_ticks: int = 500, eps_start: float = 0.2, eps_end: float = 0.0, m: int = 3):
    """SIS network dynamics dx_i = [-d x_i + eps(t) (1 - x_i) sum_j A_ij x_j] dt + eta dW_i (reference
    dataset/spdata_sde_SIS_dynamic_gene.py:44-52; d = 0.5, eta = 1e-4 as in paper_figures/common.py:45), the infection rate eps(t)
    ramped down through the epidemic threshold.  -> record dict like `biomass_trajectory`."""
    rng = np.random.default_rng(seed)
    A = ba_graph(nodes, m=m, seed=seed)
    x0 = rng.uniform(0.2, 0.8, nodes)
    ys, ramp = _euler_maruyama(lambda x, e: -d * x + e * (1 - x) * (A @ x), x0, ticks, burn_ticks, dt, eta, eps_start, eps_end, rng)
    return {"ys_dynamic": torch.from_numpy(ys), "ts_dynamic": torch.arange(ticks, dtype=torch.float32) * dt,
            "tp_values": torch.from_numpy(ramp.astype(np.float32))}


def neuronal_trajectory(nodes: int = 500, ticks: int = 100_000, seed: int = 0, tau: float = 2.0, mu: float = 1.5, eta: float = 0.01,
                        dt: float = 0.1, burn_ticks: int = 500, eps_start: float = 0.0, eps_end: float = 0.9, p: float = 0.02):
    """Wilson-Cowan network dynamics dx_i = [-x_i + (1 - eps(t)) sum_j A_ij sigmoid(tau (x_j - mu))] dt + eta dW_i (reference
    dataset/spdata_sde_neuronal_dynamic_gene.py:50-63; tau = 2, mu = 1.5, eta = 0.01) on an Erdos-Renyi graph."""
    rng = np.random.default_rng(seed)
    A = er_graph(nodes, p=p, seed=seed)
    x0 = rng.uniform(2.0, 6.0, nodes)
    ys, ramp = _euler_maruyama(lambda x, e: -x + (1 - e) * (A @ (1.0 / (1.0 + np.exp(-tau * (x - mu))))), x0, ticks, burn_ticks, dt, eta,
                               eps_start, eps_end, rng)
    return {"ys_dynamic": torch.from_numpy(ys), "ts_dynamic": torch.arange(ticks, dtype=torch.float32) * dt,
            "tp_values": torch.from_numpy(ramp.astype(np.float32))}


def sis_trajectory_device(ctx, nodes: int = 100, ticks: int = 10_000, seed: int = 0, sampling_stride: int = 1, d: float = 0.5, eta: float = 1e-4,
                          dt: float = 0.1, burn_ticks: int = 500, eps_start: float = 0.2, eps_end: float = 0.0, m: int = 3):
    """`sis_trajectory` integrated by the HIP kernel (csrc/biomass_sim.hip, model SIS) -> (series [V, L, 1] on ctx.device, time stamps)."""
    row_ptr, col_idx = adjacency_csr(ba_graph(nodes, m=m, seed=seed))
    x0 = np.random.default_rng(seed).uniform(0.2, 0.8, nodes).astype(np.float32)
    series = ctx.network_sde_simulate("SIS", torch.from_numpy(row_ptr).to(ctx.device), torch.from_numpy(col_idx).to(ctx.device),
                                      torch.from_numpy(x0).to(ctx.device).unsqueeze(0), ticks, burn_ticks, sampling_stride, 0.0, 0.0, d, eta, dt,
                                      eps_start, eps_end, seed)
    return series[0], (np.arange(ticks, dtype=np.float32) * np.float32(dt))[::sampling_stride]


def neuronal_trajectory_device(ctx, nodes: int = 500, ticks: int = 100_000, seed: int = 0, sampling_stride: int = 1, tau: float = 2.0, mu: float = 1.5,
                               eta: float = 0.01, dt: float = 0.1, burn_ticks: int = 500, eps_start: float = 0.0, eps_end: float = 0.9, p: float = 0.02):
    """`neuronal_trajectory` (Wilson-Cowan on an Erdos-Renyi graph) integrated by the HIP kernel."""
    row_ptr, col_idx = adjacency_csr(er_graph(nodes, p=p, seed=seed))
    x0 = np.random.default_rng(seed).uniform(2.0, 6.0, nodes).astype(np.float32)
    series = ctx.network_sde_simulate("neuronal", torch.from_numpy(row_ptr).to(ctx.device), torch.from_numpy(col_idx).to(ctx.device),
                                      torch.from_numpy(x0).to(ctx.device).unsqueeze(0), ticks, burn_ticks, sampling_stride, tau, mu, 0.0, eta, dt,
                                      eps_start, eps_end, seed)
    return series[0], (np.arange(ticks, dtype=np.float32) * np.float32(dt))[::sampling_stride]


def slbp_series_device(ctx, ticks: int = 100_000, seed: int = 0, sampling_stride: int = 100, dt: float = 0.1, burn_ticks: int = 2000,
                       n_start: float = 0.0, n_end: float = 3.0, **params):
    """The 2-feature SLBP series (bream, pike) under a nutrient ramp, integrated by the HIP kernel (csrc/biomass_sim.hip::slbp_em_kernel;
    reference dataset/SLBP_dynamic_gene.py) -> (series [L, 2] on ctx.device, time stamps, nutrient level per stamp)."""
    x0 = torch.from_numpy(np.random.default_rng(seed).uniform(0.0, 1.0, (1, 2)).astype(np.float32)).to(ctx.device)   # torch.rand(2, 1) there
    series = ctx.slbp_simulate(x0, ticks, burn_ticks, sampling_stride, dt, n_start, n_end, seed, **params)[0]
    t = (np.arange(ticks, dtype=np.float32) * np.float32(dt))[::sampling_stride]
    n = (np.float32(n_start) + np.arange(ticks, dtype=np.float32) * np.float32((n_end - n_start) / max(ticks - 1, 1)))[::sampling_stride]
    return series, t, n
