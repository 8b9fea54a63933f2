"""SLBP figure-5/6 cache analyses of the harness mirror against vectors recorded from the reference's own functions
(tests/golden/make_golden_slbp.py).  gx-side logic runs on CPU; anything that summarises a sampling cache goes
through the HIP variance reduction and is marked gpu."""
from pathlib import Path

import numpy as np
import pytest
import torch
import yaml

GOLD = Path(__file__).resolve().parent / "golden" / "slbp_fig_cases.npz"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture()
def H():
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as harness
    return harness


def sweep_root(tmp_path, gold):
    windows, pred_len, sampling_t, step, n = (int(v) for v in gold["dims"])
    root = tmp_path / "sweep"
    (root / "models").mkdir(parents=True)
    (root / "datas").mkdir()
    with open(root / "models" / "dataset__w40.yaml", "w") as f:
        yaml.safe_dump({"dataset": {"windows": windows, "pred_len": pred_len, "sampling_t": sampling_t}}, f)
    return root, step


def caches(gold):
    return [torch.from_numpy(x) for x in gold["sampling_cache"]], [torch.from_numpy(x) for x in gold["gx_cache"]]


def test_cache_classification_and_gx_summary(H, gold):
    sampling, gx = caches(gold)
    assert H._slbp_cache_elements_have_ndim(sampling, 3) and not H._slbp_cache_elements_are_gx(sampling)
    assert H._slbp_cache_elements_are_gx(gx) and H._slbp_cache_elements_are_gx([g.unsqueeze(0) for g in gx])
    assert not H._slbp_cache_elements_have_ndim([], 3) and not H._slbp_cache_elements_are_gx([])
    np.testing.assert_allclose(H.summarize_slbp_gx_for_fig6(gx, pred_dim=0), gold["fig6_gx_mpv"], rtol=1e-6)
    np.testing.assert_allclose(H.summarize_slbp_gx_for_fig6([g.unsqueeze(0) for g in gx], pred_dim=0), gold["fig6_gx_mpv"], rtol=1e-6)
    with pytest.raises(IndexError):
        H.summarize_slbp_gx_for_fig6(gx, pred_dim=2)
    with pytest.raises(ValueError):
        H.summarize_slbp_gx_for_fig6(sampling)
    with pytest.raises(ValueError, match="Unsupported SLBP MPV cache"):
        H.summarize_slbp_mpv_cache_for_fig5([torch.zeros(3)])
    assert H.summarize_slbp_mpv_cache_for_fig5(gx, pred_dim=0)[1] == "gx"
    assert H._legacy_single_underscore_model_name("dataset__w40_x__y") == "dataset_w40_x__y"


def test_raw_window_variance(H, gold):
    windows, pred_len, sampling_t, step, n = (int(v) for v in gold["dims"])
    r = H.slbp_raw_window_variance(torch.from_numpy(gold["series"]), gold["time"], windows=windows, sampling_t=sampling_t,
                                   sample_window_step=step, pred_dim=1)
    np.testing.assert_allclose(r["variance"], gold["raw_var"], rtol=1e-6)
    assert np.array_equal(np.asarray(r["time_points"]), gold["raw_var_tp"])
    with pytest.raises(IndexError):
        H.slbp_raw_window_variance(torch.from_numpy(gold["series"]), gold["time"], pred_dim=5)
    with pytest.raises(ValueError):
        H.slbp_raw_window_variance(torch.zeros(5000), gold["time"][:5000])


def test_gx_analysis_finds_legacy_named_shared_cache(H, gold, tmp_path):
    """No model is needed: the gx cache is found under the legacy single-underscore name of the shared file."""
    root, step = sweep_root(tmp_path, gold)
    _, gx = caches(gold)
    H._save_tensor_list(gx, H.slbp_fig6_cache_path(root, "dataset_w40", "increase", step, None, "pred_future"))
    r = H.slbp_gx_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], data_trend="increase",
                           pred_dim=0, sample_window_step=step)
    np.testing.assert_allclose(r["gx_mpv"], gold["gxa_mpv"], rtol=1e-6)
    assert np.array_equal(np.asarray(r["time_points"]), gold["gxa_tp"])
    assert Path(r["cache_path"]).name == str(gold["gxa_cache_name"])
    assert r["windows"] == 40 and r["sample_window_step"] == step
    # an old-kind cache that is not a gx cache is an error, as in the reference
    sampling, _ = caches(gold)
    H._save_tensor_list(sampling, H.slbp_fig6_cache_path(root, "dataset__w40", "increase", step, None, "gx"))
    with pytest.raises(ValueError, match="not a gx cache"):
        H.slbp_gx_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], sample_window_step=step)


def test_fig5_gx_cache_infers_window_step(H, gold, tmp_path):
    root, step = sweep_root(tmp_path, gold)
    _, gx = caches(gold)
    cp = root / "datas" / "fig5_gx.pt"
    H._save_tensor_list(gx[:-5], cp)
    r = H.slbp_mpv_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], cp, pred_dim=0, sample_window_step=3)
    np.testing.assert_allclose(r["mpv"], gold["fig5_gx_mpv"], rtol=1e-6)
    assert np.array_equal(np.asarray(r["time_points"]), gold["fig5_gx_tp"])
    assert r["sample_window_step"] == int(gold["fig5_gx_step"]) and r["uncertainty_source"] == "gx"


def test_sampling_analysis_reports_unavailable_without_model(H, gold, tmp_path):
    """Nothing cached and no checkpoint: allow_unavailable turns the failure into a result record."""
    root, step = sweep_root(tmp_path, gold)
    r = H.slbp_sampling_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], sample_window_step=step)
    assert r["available"] is False and r["mpv"] == [] and "not found" in r["reason"]
    with pytest.raises(FileNotFoundError):
        H.slbp_sampling_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], sample_window_step=step,
                                 allow_unavailable=False)
    with pytest.raises(ValueError, match="cache_kind"):
        (root / "model_dir").mkdir()
        with open(root / "model_dir" / "model_trained.yaml", "w") as f:
            yaml.safe_dump({"dataset": {"windows": 40, "pred_len": 20, "sampling_t": 10}, "task_model": "NsDiff_model_variants"}, f)
        H.slbp_direct_model_cache_analysis(root / "model_dir", torch.from_numpy(gold["series"]), gold["time"],
                                           root / "datas" / "none.pt")


@pytest.mark.gpu
def test_sampling_analysis_and_fig5_sampling_cache(H, gold, tmp_path):
    root, step = sweep_root(tmp_path, gold)
    sampling, gx = caches(gold)
    mpv, dim = H.summarize_slbp_sampling_for_fig6(sampling, pred_dim=1)
    np.testing.assert_allclose(mpv, gold["fig6_sampling_mpv"], rtol=1e-5)
    assert np.array_equal(np.asarray(dim, dtype=np.float64), gold["fig6_sampling_dim"])
    H._save_tensor_list(gx, H.slbp_fig6_cache_path(root, "dataset__w40", "increase", step, None, "pred_future"))
    H._save_tensor_list(sampling, H.slbp_fig6_cache_path(root, "dataset__w40", "increase", step, None, "sampling_pred_future"))
    r = H.slbp_sampling_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], data_trend="increase",
                                 pred_dim=1, sample_window_step=step)
    assert r["available"], r["reason"]
    np.testing.assert_allclose(r["mpv"], gold["sa_mpv"], rtol=1e-5)
    assert np.array_equal(np.asarray(r["intrinsic_dimension"], dtype=np.float64), gold["sa_dim"])
    assert np.array_equal(np.asarray(r["time_points"]), gold["sa_tp"]) and Path(r["cache_path"]).name == str(gold["sa_cache_name"])
    cp = root / "datas" / "fig5_sampling.pt"
    H._save_tensor_list(sampling[:-3], cp)
    r = H.slbp_mpv_analysis(root, "dataset__w40", torch.from_numpy(gold["series"]), gold["time"], cp, pred_dim=0, sample_window_step=3)
    np.testing.assert_allclose(r["mpv"], gold["fig5_sampling_mpv"], rtol=1e-5)
    assert np.array_equal(np.asarray(r["time_points"]), gold["fig5_sampling_tp"])
    assert r["sample_window_step"] == int(gold["fig5_sampling_step"]) and r["uncertainty_source"] == "sampling"
    # direct-model variant on the same cache (no model load needed while the cache exists)
    (root / "model_dir").mkdir()
    with open(root / "model_dir" / "model_trained.yaml", "w") as f:
        yaml.safe_dump({"dataset": {"windows": 40, "pred_len": 20, "sampling_t": 10}, "task_model": "NsDiff_model_variants"}, f)
    d = H.slbp_direct_model_cache_analysis(root / "model_dir", torch.from_numpy(gold["series"]), gold["time"], cp, pred_dim=0,
                                           sample_window_step=3)
    np.testing.assert_allclose(d["mpv"], gold["fig5_sampling_mpv"], rtol=1e-5)
    assert np.array_equal(np.asarray(d["time_points"]), gold["fig5_sampling_tp"])
