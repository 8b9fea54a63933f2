"""CPU restatement of the on-device biomass Euler-Maruyama generator (TEST INFRASTRUCTURE ONLY).

Follows csrc/biomass_sim.hip operation by operation in float32 (drift terms of the reference simulator
dataset/spdata_sde_biomass_dynamic_gene.py:53-68, noise from oracle/philox.py with the device keying), so the HIP
trajectory can be compared tick by tick.  Parity status: UNPINNED against the reference simulator (it integrates with
the absent torchsde package and a different random stream); pinned only as "the same arithmetic as the kernel".
"""
import numpy as np

from . import philox as PH


def csr(adj: np.ndarray):
    rows, cols = np.nonzero(adj)
    row_ptr = np.zeros(adj.shape[0] + 1, dtype=np.int32)
    np.add.at(row_ptr, rows + 1, 1)
    return np.cumsum(row_ptr).astype(np.int32), cols.astype(np.int32)


def simulate(adj: np.ndarray, x0: np.ndarray, ticks: int, burn_ticks: int, stride: int, r, k, d, eta, dt, eps_start, eps_end,
             seed: int) -> np.ndarray:
    """-> [V, ceil(ticks/stride)] float32, trajectory 0 of mpv_biomass_simulate."""
    f = np.float32
    V = adj.shape[0]
    row_ptr, col = csr(adj)
    total = burn_ticks + ticks
    z = PH.normals(seed, 0, V, total, 1)[:, :, 0]                       # [V, total]
    x = x0.astype(f).copy()
    xc = np.maximum(x, f(0))
    out = np.empty((V, (ticks + stride - 1) // stride), dtype=f)
    eps_step = f((f(eps_end) - f(eps_start)) / f(ticks - 1)) if ticks > 1 else f(0)
    noise_scale = f(f(eta) * np.sqrt(f(dt)))
    deg = (row_ptr[1:] - row_ptr[:-1]).astype(f)
    r, k, d, dt = f(r), f(k), f(d), f(dt)
    for t in range(total):
        i = t - burn_ticks
        eps = f(eps_start) if i < 0 else f(f(eps_start) + f(f(i) * eps_step))
        ax = np.zeros(V, dtype=f)
        for n in range(V):
            s = f(0)
            for j in range(row_ptr[n], row_ptr[n + 1]):
                s = f(s + xc[col[j]])
            ax[n] = s
        logistic = f(r * xc) * (f(1) - xc / k)
        harvest = f(eps * xc) * xc / (xc * xc + f(1))
        drift = (logistic - harvest) + d * (ax - deg * xc)
        x = (x + drift * dt) + noise_scale * z[:, t]
        xc = np.maximum(x, f(0))
        if i >= 0 and i % stride == 0:
            out[:, i // stride] = x
    return out


def simulate_model(model: str, adj: np.ndarray, x0: np.ndarray, ticks: int, burn_ticks: int, stride: int, p0, p1, p2, eta, dt, eps_start, eps_end,
                   seed: int) -> np.ndarray:
    """mpv_network_sde_simulate, trajectory 0, operation by operation in float32: 'SIS' (p2 = d; reference
    dataset/spdata_sde_SIS_dynamic_gene.py:44-52) and 'neuronal' (p0 = tau, p1 = mu; dataset/spdata_sde_neuronal_dynamic_gene.py:50-63).
    UNPINNED against the reference simulators (torchsde absent), like `simulate`."""
    if model == "biomass":
        return simulate(adj, x0, ticks, burn_ticks, stride, p0, p1, p2, eta, dt, eps_start, eps_end, seed)
    f = np.float32
    V = adj.shape[0]
    row_ptr, col = csr(adj)
    total = burn_ticks + ticks
    z = PH.normals(seed, 0, V, total, 1)[:, :, 0]
    x = x0.astype(f).copy()
    sig = lambda v: (f(1) / (f(1) + np.exp(-f(p0) * (v - f(p1)), dtype=f))).astype(f)
    share = sig(x) if model == "neuronal" else x.copy()
    out = np.empty((V, (ticks + stride - 1) // stride), dtype=f)
    eps_step = f((f(eps_end) - f(eps_start)) / f(ticks - 1)) if ticks > 1 else f(0)
    noise_scale = f(f(eta) * np.sqrt(f(dt)))
    d, dt = f(p2), f(dt)
    for t in range(total):
        i = t - burn_ticks
        eps = f(eps_start) if i < 0 else f(f(eps_start) + f(f(i) * eps_step))
        ax = np.zeros(V, dtype=f)
        for n in range(V):
            s = f(0)
            for j in range(row_ptr[n], row_ptr[n + 1]):
                s = f(s + share[col[j]])
            ax[n] = s
        if model == "SIS":
            drift = f(-d) * x + (eps * (f(1) - x)) * ax
        else:
            drift = -x + (f(1) - eps) * ax
        x = (x + drift * dt) + noise_scale * z[:, t]
        share = sig(x) if model == "neuronal" else x.copy()
        if i >= 0 and i % stride == 0:
            out[:, i // stride] = x
    return out


SLBP_PARAMS = dict(i_b=3e-4, i_p=3e-4, r=7.5e-3, H1=0.5, H2=0.1, H3=20.0, H4=15.0, c_b=7.5e-5, c_p=2.75e-4, p_r=5e-2, c_e=0.1, m_p=2.25e-3, K=1.0, D=1e-5)


def simulate_slbp(x0: np.ndarray, ticks: int, burn_ticks: int, stride: int, dt, n_start, n_end, seed: int, traj: int = 0, **params) -> np.ndarray:
    """mpv_slbp_simulate, one trajectory, operation by operation in float32: drift of reference dataset/SLBP_dynamic_gene.py:34-53, diffusion
    :56-58 (additive, so Milstein = Euler-Maruyama).  x0 [2] -> [ceil(ticks/stride), 2].  UNPINNED against torchsde (absent)."""
    f = np.float32
    p = {k: f(v) for k, v in {**SLBP_PARAMS, **params}.items()}
    total = burn_ticks + ticks
    z = PH.normals(seed, 0, traj + 1, (total + 1) // 2, 4)[traj]                    # [counter][4]: two ticks per Philox block
    X, Y = f(x0[0]), f(x0[1])
    h3, h4 = f(p["H3"] * p["H3"]), f(p["H4"] * p["H4"])
    kh3 = f(p["K"] * h3)
    n_step = f((f(n_end) - f(n_start)) / f(ticks - 1)) if ticks > 1 else f(0)
    sx = f(np.sqrt(f(f(2.0) * p["D"])) * np.sqrt(f(dt)))
    sy = f(np.sqrt(f(f(0.2) * p["D"])) * np.sqrt(f(dt)))
    dt = f(dt)
    out = np.empty(((ticks + stride - 1) // stride, 2), dtype=f)
    for t in range(total):
        i = t - burn_ticks
        N = f(n_start) if i < 0 else f(f(n_start) + f(f(i) * n_step))
        z1, z2 = (z[t >> 1][2], z[t >> 1][3]) if t & 1 else (z[t >> 1][0], z[t >> 1][1])
        x2 = f(X * X)
        V = f(kh3 / f(x2 + h3))
        FR = f(x2 / f(x2 + h4))
        dX = f(f(f(p["i_b"] + f(f(p["r"] * X) * f(N / f(N + p["H1"])))) - f(p["c_b"] * x2)) - f(f(p["p_r"] * FR) * Y))
        dY = f(f(f(p["i_p"] + f(f(f(f(p["c_e"] * p["p_r"]) * FR) * Y) * f(V / f(V + p["H2"])))) - f(p["m_p"] * Y)) - f(p["c_p"] * f(Y * Y)))
        X = f(f(X + f(dX * dt)) + f(sx * z1))
        Y = f(f(Y + f(dY * dt)) + f(sy * z2))
        if i >= 0 and i % stride == 0:
            out[i // stride] = (X, Y)
    return out
