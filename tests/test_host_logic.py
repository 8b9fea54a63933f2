"""CPU tests of the host side: ABI surface, window arithmetic, loader, caches, config errors."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

from oracle import nsdiff_oracle as O

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    """dlopen works without a GPU; every function include/mpv_hip.h declares is exported."""
    from mpv_amd import _native
    header = (ROOT / "include" / "mpv_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|void|const char\*)\s+(mpv_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
    lib = ctypes.CDLL(str(_native.LIB_PATH))
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.load_library().mpv_abi_version() == _native.ABI_VERSION
    assert int(re.search(r"#define MPV_ABI_VERSION (\d+)", header).group(1)) == _native.ABI_VERSION
    assert int(re.search(r"#define MPV_COEF_COLS (\d+)", header).group(1)) == _native.COEF_COLS


def test_no_cpu_fallback():
    from mpv_amd import _native
    with pytest.raises(_native.MpvError):
        _native.Context("cpu")


def test_coef_rows_reproduce_reference_step(golden_cases):
    """The per-step coefficient table + the kernel's formula == the reference's p_sample arithmetic."""
    from mpv_amd.models.Diffusion_model.NsDiff.NsDiff_net import nsdiff_tables, posterior_coef_rows
    steps = 50
    rows = posterior_coef_rows(nsdiff_tables(steps, "linear", 1e-4, 0.02))
    assert tuple(rows.shape) == (steps, 20) and rows.dtype == torch.float32
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    tb = O.schedule_tables(steps)
    y, y0, gx = (torch.from_numpy(golden_cases["step_seed50_%s" % n]) for n in ("y", "y0", "gx"))
    for t in (49, 25, 1):
        z = torch.from_numpy(golden_cases["step_seed50_t%d_z" % t])
        eps, sig = O.eps_network(sd, y, y0, gx, t)
        c = rows[t]
        lam1 = c[0] * gx - sig * c[1]
        lam2 = gx * gx * c[2] * c[3] - sig * gx * c[4]
        sy0 = (-lam1 + torch.sqrt(lam1 * lam1 - c[5] * lam2)) / c[6]
        nz = c[7] * gx + c[8] * sy0
        y0r = c[9] * (y - c[10] * y0 - eps * torch.sqrt(nz))
        s1, s2 = c[2] * gx + c[11] * sy0, c[3] * gx + c[12] * sy0
        den = c[13] * s2 + s1
        out = c[14] * s1 / den * y0r + c[15] * s2 / den * y + (c[16] * s2 + c[17] * s1) / den * y0 + torch.sqrt(sig) * z
        np.testing.assert_allclose(out.numpy(), golden_cases["step_seed50_t%d_out" % t], rtol=2e-5, atol=2e-6)


def test_window_arithmetic_matches_reference(golden_cases):
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    keys = [str(k) for k in golden_cases["stride_keys"]]
    for k, v in zip(keys, golden_cases["stride_vals"]):
        assert H.sampling_interval_from_t(None if k == "None" else float(k)) == int(v)
    for i in range(int(golden_cases["win_cases"])):
        L, w, s, cnt = (int(v) for v in golden_cases["win%d_args" % i])
        assert H.sliding_window_count(L, w, s) == cnt
        series = torch.arange(2 * L, dtype=torch.float32).reshape(2, L, 1)
        if L >= w:
            wins, tp = H.build_sliding_windows(series, np.arange(L) * 0.5, w, s)
            assert len(wins) == cnt
            assert np.array_equal(np.array([int(x[0, 0, 0]) for x in wins]), golden_cases["win%d_first" % i])
            assert np.array_equal(tp, golden_cases["win%d_time" % i])
            assert all(tuple(x.shape) == (2, w, 1) for x in wins)
        else:
            with pytest.raises(ValueError, match="shorter than windows"):
                H.build_sliding_windows(series, np.arange(L), w, s)
    with pytest.raises(ValueError, match="Node_num"):
        H.build_sliding_windows(torch.zeros(4, 5), np.arange(5), 2, 1)


def test_slbp_windows_and_time_points(golden_cases):
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    series, time = torch.from_numpy(golden_cases["slbp_series"]), torch.from_numpy(golden_cases["slbp_time"])
    ins, outs, tp = H.build_slbp_sensitivity_windows(series, time, 200, 200, 10, 10)
    assert len(ins) == (500 - 200) // 10 + 1 and len(outs) == (300 - 200) // 10 + 1
    assert np.array_equal(tp[:8], golden_cases["slbp_tp"])
    sub = series[::100]
    assert torch.equal(ins[3], sub[30:230]) and torch.equal(outs[2], sub[200 + 20: 200 + 220])


def test_infer_window_step_from_cache_length():
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    for step in (1, 3, 5, 7, 50):
        n = H.sliding_window_count(1000, 100, step)
        got = H.infer_sample_window_step_from_cache(1000, 100, n, fallback_step=5)
        assert H.sliding_window_count(1000, 100, got) == n
    assert H.infer_sample_window_step_from_cache(1000, 100, 0, 5) == 5
    assert H.infer_sample_window_step_from_cache(50, 100, 3, 5) == 5
    assert H.infer_sample_window_step_from_cache(1000, 100, 1, 9) == 9


def test_time_series_layouts_and_record_errors(tmp_path):
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    x = torch.arange(12.0).reshape(4, 3)                           # [T, Node] for network dynamics
    assert tuple(H.normalize_time_series(x, "biomass").shape) == (3, 4, 1)
    assert tuple(H.normalize_time_series(x, "SLBP").shape) == (1, 4, 3)
    assert tuple(H.normalize_time_series(torch.zeros(2, 5, 1)).shape) == (2, 5, 1)
    with pytest.raises(ValueError):
        H.normalize_time_series(torch.zeros(3))
    torch.save({"ys_dynamic": x}, tmp_path / "biomass_bad.pt")
    with pytest.raises(KeyError, match="ys_dynamic"):
        H.load_dynamic_data(tmp_path / "biomass_bad.pt")
    torch.save({"ys_dynamic": x, "ts_dynamic": torch.arange(4.0)}, tmp_path / "biomass_x.pt")
    rec = H.load_dynamic_data(tmp_path / "biomass_x.pt")
    assert rec["dynamic_type"] == "biomass" and tuple(rec["torch_time_series"].shape) == (3, 4, 1)
    torch.save({"ys_dynamic": x, "ts_dynamic": torch.arange(4.0), "N_values": x}, tmp_path / "anything.pt")
    assert H.load_dynamic_data(tmp_path / "anything.pt")["dynamic_type"] == "SLBP"


def test_cache_roundtrip_and_paths(tmp_path):
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    items = [torch.randn(3, 4, 1, 5), torch.randn(3, 4, 1, 5)]
    H._save_tensor_list(items, tmp_path / "sub" / "c.pt")
    back = H._load_tensor_list(tmp_path / "sub" / "c.pt")
    assert isinstance(back, list) and all(torch.equal(a, b) for a, b in zip(items, back))
    assert isinstance(torch.load(tmp_path / "sub" / "c.pt"), list)                  # the reference's format
    torch.save({"not": "a list"}, tmp_path / "bad.pt")
    with pytest.raises(TypeError):
        H._load_tensor_list(tmp_path / "bad.pt")
    assert H.read_sensitivity_pred_future_cache(tmp_path / "bad.pt") is None
    assert H.data_cache_name("a/b/run7.pt", "_gx") == "run7_gx.pt" and H.data_cache_name(None) == "data.pt"
    assert H.resolve_cache_path(None, tmp_path, "x/traj.pt", "biomass") == tmp_path / "traj.pt"
    assert H.resolve_cache_path(tmp_path / "k.pt", None, None, None) == tmp_path / "k.pt"
    assert H.resolve_figure_path(tmp_path / "k.pt") == tmp_path / "k.png"


from helpers import WOFX_NET, write_model_dir as _write_model_dir  # noqa: E402


def test_loader_reads_reference_checkpoint_format(tmp_path, ckpt_wofx):
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd import _native
    _write_model_dir(tmp_path / "m", ckpt_wofx, WOFX_NET)
    model, net_param = H.load_model_from_dir(tmp_path / "m", device=torch.device("cpu"),
                                             infer_params={"n_z_samples": 4, "parallel_sample": 2})
    assert not model.training and net_param["n_z_samples"] == 4 and model.configs.parallel_sample == 2
    assert model.scaler == "StandardScaler" and torch.equal(model.scaler_mean, ckpt_wofx["scaler_mean"])
    assert set(model.state_dict().keys()) == set(ckpt_wofx.keys())
    x = torch.randn(3, 200, 2)
    assert torch.allclose(model.scaler_inverse_transform(model.scaler_transform(x)), x, atol=1e-5)
    with pytest.raises(_native.MpvError, match="no CPU fallback"):              # fails loudly off-GPU
        model.evaluation_step(x)
    with pytest.raises(_native.MpvError):
        model.cond_pred_model_g(x)
    with pytest.raises(FileNotFoundError):
        H.read_model_config(tmp_path / "nope")
    # a 'module.'-prefixed (DataParallel) checkpoint loads too
    st = torch.load(tmp_path / "m" / "model_trained")
    st["state_dict"] = {"module." + k: v for k, v in st["state_dict"].items()}
    torch.save(st, tmp_path / "m" / "model_trained")
    model2, _ = H.load_model_from_dir(tmp_path / "m", device=torch.device("cpu"))
    assert torch.equal(model2.model.diffussion_model.lin2.lin.weight, ckpt_wofx["model.diffussion_model.lin2.lin.weight"])


def test_factory_and_config_errors(ckpt_wofx):
    from mpv_amd.models.models import diffusion_models
    with pytest.raises(ValueError, match="don't exit"):
        diffusion_models("nope", {})
    with pytest.raises(KeyError):                                                   # NsDiff_spatial is built (tests/test_nsdiff_spatial_oracle.py);
        diffusion_models("NsDiff_spatial", {})                                      # like the reference it needs train_model_select
    dts = diffusion_models("DiffusionTS", dict(device="cpu", dataset_nf=1, windows=12, pred_len=8, d_model=16, n_layer_enc=1, n_layer_dec=1,
                                               n_heads=2, timesteps=20, diffusion_steps=5, n_z_samples=5, parallel_sample=2))
    with pytest.raises(ValueError, match="divisible by parallel_sample"):          # reference DiffusionTS_model.py:83-84
        dts.evaluation_step(torch.zeros(1, 12, 1))
    with pytest.raises(Exception, match="ROCm device|no CPU fallback"):             # the product path never computes on the CPU
        diffusion_models("DiffusionTS", dict(device="cpu", dataset_nf=1, windows=12, pred_len=8, d_model=16, n_layer_enc=1, n_layer_dec=1,
                                             n_heads=2, timesteps=20, diffusion_steps=5, n_z_samples=4, parallel_sample=2)
                         ).evaluation_step(torch.zeros(1, 12, 1))
    base = dict(WOFX_NET, device="cpu")
    with pytest.raises(ValueError):
        diffusion_models("NsDiff_model_variants", dict(base), train_model_select="bogus")
    cm = diffusion_models("NsDiff_model_variants", dict(base), train_model_select="cond_mean")     # constructs; f(x) is attached later
    assert cm.needs_backbone and cm.cond_pred_model is None and cm.cond_pred_model_g is None
    with pytest.raises(NotImplementedError, match="f\\(x\\)"):
        cm.evaluation_step(torch.zeros(1, 200, 2))
    with pytest.raises(ValueError, match="rolling_length"):
        diffusion_models("NsDiff_model_variants", dict(base, rolling_length=200), train_model_select="cond_var")
    g = diffusion_models("NsDiff", dict(base), train_model_select="pretrain_g")       # shipped NsDiff_preg layout
    assert set(g.state_dict().keys()) == {k for k in ckpt_wofx if not k.startswith("model.")}
    m = diffusion_models("NsDiff_model_variants", dict(base, n_z_samples=10, parallel_sample=4), train_model_select="Guassian")
    assert m.cond_pred_model_g is None
    with pytest.raises(AssertionError, match="multiple of parallel_sample"):
        m._check_sampling_config()


def test_uncertainty_ews_gx_from_cache_and_argument_errors(tmp_path, ckpt_wofx, golden_cases):
    """Config / cache / result-dict plumbing of uncertainty_ews without a GPU (gx cache present)."""
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    sd = O.random_state_dict(1, 50, 100, 100, 50, seed=123)
    net = dict(WOFX_NET, diffusion_steps=50, rolling_length=50, windows=100, pred_len=100, dataset_nf=1)
    mdir = tmp_path / "model"
    _write_model_dir(mdir, sd, net)
    series = torch.from_numpy(np.ascontiguousarray(golden_cases["net_series"]))[:, :, 0].t().contiguous()   # [ticks, V]
    rec = {"ys_dynamic": series, "ts_dynamic": torch.arange(series.shape[0]) * 0.1, "tp_values": torch.zeros(3)}
    torch.save(rec, tmp_path / "biomass_run.pt")
    gx_list = [torch.from_numpy(g) for g in golden_cases["net_gx"]]
    H._save_tensor_list(gx_list, mdir / "biomass_run_gx.pt")
    res = H.uncertainty_ews(model_save_file=mdir, data_file=tmp_path / "biomass_run.pt", dynamic_type="biomass",
                            sampling_t=0.1, sample_window_step=80, uncertainty_method="preg", device=torch.device("cpu"))
    assert res["uncertainty_source"] == "gx" and res["sample_window_step"] == 80
    auto = H.uncertainty_ews(model_save_file=mdir, data_file=tmp_path / "biomass_run.pt", dynamic_type="biomass",
                             sampling_t=0.1, uncertainty_method="gx", device=torch.device("cpu"))
    assert auto["sample_window_step"] == 54          # inferred from the cache length exactly as the reference does
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float32), golden_cases["net_g_ews"], rtol=1e-6)
    np.testing.assert_allclose(res["time_points"], golden_cases["net_tp"])
    assert res["windows"] == 100 and res["pred_len"] == 100 and res["task_model"] == "NsDiff_model_variants"
    assert set(res) >= {"pred_future_list", "pred_mean", "ews", "time_points", "cache_path", "figure_path",
                        "torch_time_series", "time_data", "dynamic_type", "sampling_t", "sample_window_step", "windows",
                        "pred_len", "task_model", "uncertainty_method", "uncertainty_source", "graph_file",
                        "model_save_file", "loaded_net_param", "nsdiff_g"}
    with pytest.raises(ValueError, match="uncertainty_method"):
        H.uncertainty_ews(model_save_file=mdir, data_file=tmp_path / "biomass_run.pt", uncertainty_method="bogus")
    with pytest.raises(ValueError, match="Provide data_file"):
        H.uncertainty_ews(model_save_file=mdir)
    with pytest.raises(ValueError, match="time_data is required"):
        H.uncertainty_ews(model_save_file=mdir, torch_time_series=series)
    with pytest.raises(ValueError, match="dataset.windows"):
        H.uncertainty_ews(torch_time_series=series, time_data=rec["ts_dynamic"])


def test_synthetic_inputs_are_seeded_and_shaped():
    from mpv_amd import synthetic
    a = synthetic.biomass_trajectory(nodes=12, ticks=300, seed=3, burn_ticks=50)
    b = synthetic.biomass_trajectory(nodes=12, ticks=300, seed=3, burn_ticks=50)
    assert torch.equal(a["ys_dynamic"], b["ys_dynamic"]) and tuple(a["ys_dynamic"].shape) == (300, 12)
    adj = synthetic.ba_graph(30, 3, seed=1)
    assert np.array_equal(adj, adj.T) and adj.diagonal().sum() == 0 and (adj.sum(1) >= 3).all()


def test_backbone_plumbing_without_gpu(ckpt_wofx):
    """f(x)-using selects construct, load a checkpoint that carries backbone weights, and refuse to sample until a
    backbone or y0hat is given; with one attached, evaluation reaches the native call (which has no CPU fallback)."""
    from mpv_amd import _native
    from mpv_amd.models.models import diffusion_models
    net = dict(device="cpu", scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=100,
               diffusion_schedule="linear", diffusion_steps=20, beta_start=1e-4, beta_end=0.02, n_z_samples=4, parallel_sample=2,
               windows=200, pred_len=200, dataset_nf=2)
    sd = dict(ckpt_wofx)
    sd["cond_pred_model.projection.weight"] = torch.zeros(2, 8)               # what a full checkpoint also holds
    batch = torch.zeros(3, 200, 2)
    for select, cls_task in (("wo_UANS", "NsDiff_model_variants"), ("cond_mean", "NsDiff_model_variants"), ("NsDiff_model", "NsDiff")):
        m = diffusion_models(cls_task, dict(net, load_pretrain=False), train_model_select=select)
        own = set(m.state_dict())
        m.load_state_dict({k: v for k, v in sd.items() if k in own or k.startswith("cond_pred_model.")})
        assert m.needs_backbone and m.cond_pred_model is None
        with pytest.raises(NotImplementedError, match="attach_backbone"):
            m.evaluation_step(batch)

        class Stub(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.projection = torch.nn.Linear(8, 2, bias=False)
                self.calls = []

            def forward(self, x, dec_inp):
                self.calls.append((tuple(x.shape), tuple(dec_inp.shape)))
                return x[:, -1:, :].expand(-1, 200, -1) * 0.5, None

        stub = Stub()
        m.attach_backbone(stub)
        assert "cond_pred_model.projection.weight" in m.state_dict()
        with pytest.raises(_native.MpvError):                                    # CPU: reaches the native path
            m.evaluation_step(batch)
    with pytest.raises(NotImplementedError):
        diffusion_models("NsDiff", dict(net, load_pretrain=False), train_model_select="pretrain_f")


def test_model_hook_runs_before_the_state_dict_is_loaded(tmp_path, ckpt_wofx):
    """utils.MODEL_HOOK / model_hook=: attach a backbone first, then its checkpoint entries load strictly."""
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.utils import utils as U
    net = dict(device="cpu", scaler_type="StandardScaler", task_model="NsDiff_model_variants", rolling_length=100,
               diffusion_schedule="linear", diffusion_steps=20, beta_start=1e-4, beta_end=0.02, n_z_samples=4, parallel_sample=2,
               windows=200, pred_len=200, dataset_nf=2)

    class Stub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.projection = torch.nn.Linear(8, 2, bias=False)

    m = diffusion_models("NsDiff_model_variants", dict(net), train_model_select="wo_UANS").attach_backbone(Stub())
    with torch.no_grad():
        m.cond_pred_model.projection.weight.fill_(0.25)
    own = set(m.state_dict())
    m.load_state_dict({**{k: v for k, v in ckpt_wofx.items() if k in own}, "cond_pred_model.projection.weight": torch.full((2, 8), 0.25)})
    U.save_checkpoint(str(tmp_path), "model_trained", m, dict(net))
    seen = []
    hook = lambda model: (seen.append(type(model).__name__), model.attach_backbone(Stub()))
    got, _ = U.load_diffusion_model(str(tmp_path / "model_trained"), "cpu", train_model_select="wo_UANS", model_hook=hook)
    assert seen == ["NsDiff_model_variants"] and float(got.cond_pred_model.projection.weight.mean()) == 0.25
    U.MODEL_HOOK = hook
    try:
        got2, _ = U.load_diffusion_model(str(tmp_path / "model_trained"), "cpu", train_model_select="wo_UANS")
        assert float(got2.cond_pred_model.projection.weight.mean()) == 0.25
    finally:
        U.MODEL_HOOK = None
    plain, _ = U.load_diffusion_model(str(tmp_path / "model_trained"), "cpu", train_model_select="wo_UANS")    # no hook: entries dropped
    assert plain.cond_pred_model is None


def test_headline_sampler_kernel_keeps_its_step_body_free_of_scratch_traffic():
    """Build-time guard: the split-f16 chain kernel is vector-issue bound, so a spill inside the reverse-step body costs
    issue slots directly.  tools/check_spills.py compiles the kernel, finds the step body of every instantiation and exits 1 if the
    headline one (F = 1, Philox, three products) touches scratch there."""
    import subprocess
    import sys
    from conftest import ROOT
    run = subprocess.run([sys.executable, str(ROOT / "tools" / "check_spills.py")], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "F=1, philox=1, products=3" in run.stdout
