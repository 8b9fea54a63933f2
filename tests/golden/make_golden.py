#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE's own code.

Runs only in the build container (needs /root/reference).  Nothing here travels to the
GPU box except the .npz files it writes.  The reference is imported unmodified; the
third-party packages it imports at module level but never calls on this path
(torch_timeseries, torch_geometric, igraph, ...) are satisfied by empty
placeholder modules so that `NsDiff_net`, `SigmaEstimation` and `NsDiff_model_variants`
import.  Only code paths that execute reference code end-to-end are recorded
(`cond_var` / `Guassian` variants: no f(x) backbone).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Every normal draw the reference makes (`torch.randn_like`) is recorded, so the oracle
and the HIP path can replay the identical noise.
"""
import importlib.abc
import importlib.machinery
import os
import sys
import types
from pathlib import Path

import numpy as np
import torch

REF = Path(os.environ.get("MPV_REFERENCE", "/root/reference"))
HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))

ABSENT = ("torch_timeseries", "torch_geometric", "torch_sparse", "torch_scatter", "igraph", "torchsde")


class _Placeholder(torch.nn.Module):
    def __init__(self, *a, **k):
        super().__init__()


class _AbsentFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """Resolve `import <absent>[.sub...]` to an empty module whose attributes are inert."""

    def find_spec(self, name, path=None, target=None):
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = types.ModuleType(spec.name)
        m.__path__ = []
        m.__all__ = []
        m.__getattr__ = lambda attr: _Placeholder
        return m

    def exec_module(self, module):
        pass


def import_reference():
    sys.meta_path.append(_AbsentFinder())
    sys.path.insert(0, str(REF))
    from models.Diffusion_model.NsDiff import nsdiff_utils, denoise, sigma, g_backbone, NsDiff_net, NsDiff_model
    from models import models as factory
    from evaluation_and_analysis import diffusion_model_uncertainy as harness
    return types.SimpleNamespace(utils=nsdiff_utils, denoise=denoise, sigma=sigma, g=g_backbone,
                                 net=NsDiff_net, model=NsDiff_model, factory=factory, harness=harness)


class NoiseTap:
    """Record every torch.randn_like draw made while active."""

    def __enter__(self):
        self.draws = []
        self._orig = torch.randn_like

        def tapped(x, *a, **k):
            z = self._orig(x, *a, **k)
            self.draws.append(z.detach().clone())
            return z

        torch.randn_like = tapped
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig

    def flat(self):
        return np.concatenate([d.reshape(-1).numpy() for d in self.draws]) if self.draws else np.zeros(0, np.float32)


def npy(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


def net_param(nf, steps, windows, pred_len, rolling, n_z, par, schedule="linear"):
    return dict(device="cpu", scaler_type="StandardScaler", task_model="NsDiff_model_variants",
                rolling_length=rolling, diffusion_schedule=schedule, diffusion_steps=steps,
                beta_start=1e-4, beta_end=0.02, n_z_samples=n_z, parallel_sample=par,
                load_pretrain=False, freeze_pretrain=False, windows=windows, pred_len=pred_len,
                dataset_nf=nf)


def build_variant(ref, sd, select, **kw):
    model = ref.factory.diffusion_models("NsDiff_model_variants", net_param(**kw), train_model_select=select)
    own = model.state_dict()
    model.load_state_dict({k: sd[k] for k in own}, strict=True)
    return model.eval()


def slbp_like_series(length, seed):
    """Any seeded 2-feature series will do for plumbing."""
    rng = np.random.default_rng(seed)
    t = np.arange(length) * 0.1
    a = 2.0 + 0.4 * np.sin(t / 40.0) + np.cumsum(rng.normal(0, 0.01, length))
    b = 1.0 + 0.2 * np.cos(t / 25.0) + np.cumsum(rng.normal(0, 0.015, length)) + 0.0004 * t
    return torch.tensor(np.stack([a, b], axis=1), dtype=torch.float32), torch.tensor(t, dtype=torch.float32)


def network_series(nodes, length, seed):
    rng = np.random.default_rng(seed)
    base = 5.0 + np.cumsum(rng.normal(0, 0.02, (length, 1)), axis=0)
    x = base + 0.3 * rng.normal(0, 1, (length, nodes)) * np.linspace(0.5, 2.0, length)[:, None]
    return torch.tensor(x, dtype=torch.float32), torch.arange(length, dtype=torch.float32) * 0.1


def main():
    from oracle import nsdiff_oracle as O

    ref = import_reference()
    out = {}

    # ---- 1. schedule tables (NsDiff_net.__init__) -------------------------------------
    sched = {}
    for steps, kind in ((10, "linear"), (20, "linear"), (50, "linear"), (8, "quad"), (12, "cosine")):
        cfg = types.SimpleNamespace(diffusion_steps=steps, diffusion_schedule=kind, beta_start=1e-4,
                                    beta_end=0.02, dataset_nf=1)
        net = ref.net.NsDiff_net(cfg, "cpu")
        for name in ("betas", "alphas", "alphas_cumprod", "alphas_cumprod_sum", "alphas_hat", "betas_tilde",
                     "betas_bar", "betas_tilde_m_1", "betas_bar_m_1", "one_minus_alphas_bar_sqrt",
                     "alphas_cumprod_prev", "alphas_cumprod_sum_prev"):
            sched["%s_%d_%s" % (kind, steps, name)] = npy(getattr(net, name))
    np.savez_compressed(HERE / "nsdiff_schedule.npz", **sched)

    # ---- real checkpoints shipped with the reference (data, not source) ----------------
    for tag, rel in (("wofx", "ews_results/NsDiff_machine/wo_fx/model_trained"),
                     ("sim", "ews_results/real_data/model/simulation/model_trained")):
        st = torch.load(REF / rel, map_location="cpu", weights_only=True)
        np.savez_compressed(HERE / ("ckpt_%s.npz" % tag), **{k: npy(v) for k, v in st["state_dict"].items()})
    wofx = {k: v for k, v in torch.load(REF / "ews_results/NsDiff_machine/wo_fx/model_trained",
                                        map_location="cpu", weights_only=True)["state_dict"].items()}

    g = torch.Generator().manual_seed(7)

    # ---- 2. trailing variance (sigma.py) ----------------------------------------------
    x = torch.randn(3, 40, 2, generator=g) * 2 + 1
    out["rollvar_x"] = npy(x)
    for k in (1, 7, 40):
        out["rollvar_k%d" % k] = npy(ref.sigma.wv_sigma_trailing(x, k))

    # ---- 3. sigma head (g_backbone.SigmaEstimation) -----------------------------------
    def ref_sigma_head(sd, nf, windows, pred_len, rolling):
        m = ref.g.SigmaEstimation(windows, pred_len, nf, 512, rolling)
        m.load_state_dict({k[len("cond_pred_model_g."):]: v for k, v in sd.items() if k.startswith("cond_pred_model_g.")})
        return m.eval()

    xs = torch.randn(2, 200, 2, generator=g)
    out["ghead_wofx_x"] = npy(xs)
    with torch.no_grad():
        out["ghead_wofx_y"] = npy(ref_sigma_head(wofx, 2, 200, 200, 100)(xs))
    seeded_f1 = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    xs = torch.randn(4, 100, 1, generator=g) * 0.5
    out["ghead_seed_x"] = npy(xs)
    with torch.no_grad():
        out["ghead_seed_y"] = npy(ref_sigma_head(seeded_f1, 1, 100, 100, 50)(xs))

    # ---- 4. epsilon network at every t -------------------------------------------------
    def ref_eps_net(sd, nf, steps):
        m = ref.denoise.ConditionalGuidedModel(steps, nf)
        pre = "model.diffussion_model."
        m.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        return m.eval()

    for tag, sd, nf, steps in (("seed", seeded_f1, 1, 50), ("wofx", wofx, 2, 20)):
        y = torch.randn(5, 7, nf, generator=g)
        y0 = torch.randn(5, 7, nf, generator=g) * 0.3
        gx = torch.rand(5, 7, nf, generator=g) + 0.05
        net = ref_eps_net(sd, nf, steps)
        eps, sig = [], []
        with torch.no_grad():
            for t in range(steps):
                e, s = net(y, y0, gx, torch.full((5,), t, dtype=torch.long))
                eps.append(npy(e))
                sig.append(npy(s))
        out.update({"eps_%s_y" % tag: npy(y), "eps_%s_y0" % tag: npy(y0), "eps_%s_gx" % tag: npy(gx),
                    "eps_%s_eps" % tag: np.stack(eps), "eps_%s_sig" % tag: np.stack(sig)})

    # ---- 5/6. single steps and full loops with recorded noise --------------------------
    def ref_diffnet(sd, nf, steps):
        cfg = types.SimpleNamespace(diffusion_steps=steps, diffusion_schedule="linear", beta_start=1e-4,
                                    beta_end=0.02, dataset_nf=nf)
        net = ref.net.NsDiff_net(cfg, "cpu")
        pre = "model."
        net.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith("model.")})
        return net.eval()

    def sched_args(net):
        return (net.alphas, net.one_minus_alphas_bar_sqrt, net.alphas_cumprod, net.alphas_cumprod_sum,
                net.alphas_cumprod_prev, net.alphas_cumprod_sum_prev, net.betas_tilde, net.betas_bar,
                net.betas_tilde_m_1, net.betas_bar_m_1)

    seeded_f2_t10 = O.random_state_dict(2, 10, 200, 200, 100, seed=321)
    for tag, sd, nf, steps in (("seed50", seeded_f1, 1, 50), ("wofx20", wofx, 2, 20), ("seed10", seeded_f2_t10, 2, 10)):
        net = ref_diffnet(sd, nf, steps)
        y = torch.randn(6, 5, nf, generator=g)
        y0 = torch.randn(6, 5, nf, generator=g) * 0.2
        gx = torch.rand(6, 5, nf, generator=g) * 0.5 + 0.02
        out.update({"step_%s_y" % tag: npy(y), "step_%s_y0" % tag: npy(y0), "step_%s_gx" % tag: npy(gx)})
        with torch.no_grad():
            for t in (steps - 1, steps // 2, 1):
                torch.manual_seed(1000 + t)
                with NoiseTap() as tap:
                    nxt = ref.utils.p_sample(net, y, y0, gx, y0, t, *sched_args(net))
                out["step_%s_t%d_z" % (tag, t)] = npy(tap.draws[0])
                out["step_%s_t%d_out" % (tag, t)] = npy(nxt)
            out["step_%s_final" % tag] = npy(ref.utils.p_sample_t_1to0(net, y, y0, gx, y0, net.one_minus_alphas_bar_sqrt,
                                                                         *((net.alphas,) + sched_args(net)[2:])))
            torch.manual_seed(77)
            with NoiseTap() as tap:
                seq = ref.utils.p_sample_loop(net, y0, gx, y0, steps, *sched_args(net))
            out["loop_%s_noise" % tag] = np.stack([npy(d) for d in tap.draws])
            out["loop_%s_out" % tag] = npy(seq[-1])

    # ---- 7. evaluation_step ---------------------------------------------------------------
    series2, time2 = slbp_like_series(50000, seed=5)
    model = build_variant(ref, wofx, "cond_var", nf=2, steps=20, windows=200, pred_len=200, rolling=100, n_z=4, par=2)
    # (a) real checkpoint, one SLBP window, two chunks
    stride = ref.harness.sampling_interval_from_t(10)
    sub = series2[::stride][:200]
    batch = model.scaler_transform(sub).unsqueeze(0)
    torch.manual_seed(11)
    with torch.no_grad(), NoiseTap() as tap:
        outs, _ = model.evaluation_step(batch.clone())
    out.update({"eval_wofx_batch": npy(batch), "eval_wofx_noise": tap.flat(), "eval_wofx_out": npy(outs)})

    # (b) seeded F=1 network shape (reduced C2): V=3, windows=pred=100, rolling 50, T=50, N=8 in chunks of 4
    net_series, net_time = network_series(3, 260, seed=9)
    sd_b = dict(seeded_f1)
    flat = net_series.reshape(-1, 1)
    sd_b["scaler_mean"] = flat.mean(0)
    sd_b["scaler_std"] = flat.std(0)
    model_b = build_variant(ref, sd_b, "cond_var", nf=1, steps=50, windows=100, pred_len=100, rolling=50, n_z=8, par=4)
    series_b = net_series.t().unsqueeze(-1)                           # [V, L, 1]
    batch = model_b.scaler_transform(series_b[:, :100, :])
    torch.manual_seed(12)
    with torch.no_grad(), NoiseTap() as tap:
        outs, _ = model_b.evaluation_step(batch.clone())
    out.update({"eval_seed_batch": npy(batch), "eval_seed_noise": tap.flat(), "eval_seed_out": npy(outs),
                "eval_seed_scaler_mean": npy(sd_b["scaler_mean"]), "eval_seed_scaler_std": npy(sd_b["scaler_std"])})

    # (c) Guassian variant (no g, no f): gx = 1, y0hat = 0
    model_c = build_variant(ref, seeded_f2_t10, "Guassian", nf=2, steps=10, windows=200, pred_len=200, rolling=100, n_z=4, par=4)
    batch = torch.randn(2, 200, 2, generator=g)
    torch.manual_seed(13)
    with torch.no_grad(), NoiseTap() as tap:
        outs, _ = model_c.evaluation_step(batch.clone())
    out.update({"eval_gauss_batch": npy(batch), "eval_gauss_noise": tap.flat(), "eval_gauss_out": npy(outs)})

    # ---- 8. window arithmetic ---------------------------------------------------------
    H = ref.harness
    strides = {str(v): H.sampling_interval_from_t(v) for v in (None, 0.05, 0.1, 0.2, 0.3, 1, 10, 100, 7.7)}
    out["stride_keys"] = np.array(list(strides.keys()))
    out["stride_vals"] = np.array(list(strides.values()), dtype=np.int64)
    grid = []
    for L, w, s in ((1000, 100, 5), (1000, 100, 50), (200, 200, 10), (199, 200, 10), (2000, 200, 10), (1003, 100, 7)):
        cnt = H.sliding_window_count(L, w, s)
        if L >= w:
            series = torch.arange(2 * L, dtype=torch.float32).reshape(2, L, 1)
            wins, tp = H.build_sliding_windows(series, np.arange(L) * 0.5, w, s)
            first = np.array([int(x[0, 0, 0]) for x in wins], dtype=np.int64)
            assert len(wins) == cnt
            grid.append((L, w, s, cnt, first, np.asarray(tp)))
        else:
            grid.append((L, w, s, cnt, np.zeros(0, np.int64), np.zeros(0)))
    for i, (L, w, s, cnt, first, tp) in enumerate(grid):
        out["win%d_args" % i] = np.array([L, w, s, cnt], dtype=np.int64)
        out["win%d_first" % i] = first
        out["win%d_time" % i] = tp
    out["win_cases"] = np.array(len(grid))

    # ---- 9. harness loops + summaries ---------------------------------------------------
    # C1: SLBP single series through run_slbp_sensitivity_cache (B=1), real checkpoint, N=4
    model_s = build_variant(ref, wofx, "cond_var", nf=2, steps=20, windows=200, pred_len=200, rolling=100, n_z=4, par=4)
    inputs, targets, tpts = H.build_slbp_sensitivity_windows(series2, time2, 200, 200, 10, 10)
    inputs = inputs[:8]
    torch.manual_seed(21)
    tmp = Path("/tmp/_golden_cache_slbp.pt")
    with NoiseTap() as tap:
        preds = H.run_slbp_sensitivity_cache(model_s, inputs, tmp, torch.device("cpu"), force_recompute=True)
    mpv, err = H.summarize_slbp_sensitivity(preds, targets[:8], model=model_s, device=torch.device("cpu"), pred_dim=0)
    mpv1, _ = H.summarize_slbp_sensitivity(preds, (), model=model_s, device=torch.device("cpu"), pred_dim=1)
    out.update({"slbp_series": npy(series2), "slbp_time": npy(time2), "slbp_tp": np.asarray(tpts[:8]),
                "slbp_noise0": np.concatenate([npy(d).reshape(-1) for d in tap.draws[:20]]),
                "slbp_pred": np.stack([npy(p) for p in preds]), "slbp_mpv0": np.array(mpv, dtype=np.float32),
                "slbp_mpv1": np.array(mpv1, dtype=np.float32), "slbp_err0": np.array(err, dtype=np.float32)})

    # network: run_evaluation_cache over 3 windows, seeded F=1 model (b), summaries
    sampled, stime = H.sample_time_series(series_b, net_time, sampling_t=0.1)
    wins, tp = H.build_sliding_windows(sampled, stime, 100, 80)
    torch.manual_seed(22)
    tmp = Path("/tmp/_golden_cache_net.pt")
    with NoiseTap() as tap:
        preds = H.run_evaluation_cache(model_b, list(wins), 100, tmp, torch.device("cpu"), force_recompute=True)
    pm, ews = H.summarize_pred_future_list(preds, model=model_b)
    with torch.no_grad():
        glist = H.run_nsdiff_g_cache(model_b, list(wins), Path("/tmp/_golden_cache_g.pt"), torch.device("cpu"),
                                     force_recompute=True)
    gm, gews = H.summarize_nsdiff_g_list(glist, pred_dim=0)
    out.update({"net_series": npy(series_b), "net_tp": np.asarray(tp), "net_noise": tap.flat(),
                "net_pred": np.stack([npy(p) for p in preds]),
                "net_pred_mean": np.array(pm, dtype=np.float32), "net_ews": np.array(ews, dtype=np.float32),
                "net_gx": np.stack([npy(x) for x in glist]), "net_g_mean": np.array(gm, dtype=np.float32),
                "net_g_ews": np.array(gews, dtype=np.float32)})

    np.savez_compressed(HERE / "nsdiff_cases.npz", **out)
    for f in sorted(HERE.glob("*.npz")):
        print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
