"""GPU tests (-m gpu): uncertainty_ews end to end for the DiffSTG and DiffusionTS model families -- model directory in the
reference's layout, data record, (for DiffSTG) a graphml topology -- against the CPU oracles driven window by window with the same
torch seed (noise_source='torch_cpu' replays the reference's normal stream)."""
import numpy as np
import pytest
import torch
import yaml

from test_gpu_parity import dev, rel_err, _reseed_after

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _write_dir(path, task_model, net, sd, dataset):
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.utils.utils import save_checkpoint
    path.mkdir(parents=True, exist_ok=True)
    m = diffusion_models(task_model, dict(net, device="cpu"))
    m.load_state_dict(sd, strict=True)
    save_checkpoint(str(path), "model_trained", m, dict(net, task_model=task_model))
    (path / "model_trained.yaml").write_text(yaml.safe_dump({"dataset": dataset, "net": dict(net, task_model=task_model, device="cpu"),
                                                              "train": {"train_model_select": task_model}}))


def test_uncertainty_ews_diffstg(tmp_path, monkeypatch):
    import networkx as nx
    from conftest import GOLDEN
    from oracle import diffstg_oracle as O
    from test_diffstg_oracle import case, state
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    z = np.load(GOLDEN / "diffstg_cases.npz")
    net, V = case(z, "tiny")
    sd = state(z, "tiny", net)
    net = dict(net, noise_source="torch_cpu")
    _write_dir(tmp_path / "m", "DiffSTG", net, sd, {"windows": net["T_h"], "pred_len": net["T_p"], "sampling_t": 0.1, "interval_step": 7})
    edge = z["edge_tiny"]
    g = nx.Graph()
    g.add_nodes_from(range(V))
    g.add_edges_from([(int(a), int(b)) for a, b in edge.T if a < b])
    nx.write_graphml(g, tmp_path / "g.graphml")
    gen = torch.Generator().manual_seed(3)
    series = torch.cumsum(torch.randn(40, V, generator=gen) * 0.2, dim=0) + 1.0                 # [ticks, V]
    torch.save({"ys_dynamic": series, "ts_dynamic": torch.arange(40) * 0.1, "tp_values": torch.zeros(1)}, tmp_path / "biomass_x.pt")
    _reseed_after(monkeypatch, H, "load_model_from_dir", 12)
    res = H.uncertainty_ews(model_save_file=tmp_path / "m", data_file=tmp_path / "biomass_x.pt", dynamic_type="biomass",
                            graph_file=tmp_path / "g.graphml", device=dev(), save_nsdiff_g=False,
                            infer_params={"noise_source": "torch_cpu"})
    n_w = (40 - net["T_h"]) // 7 + 1
    assert res["sample_window_step"] == 7 and len(res["ews"]) == n_w and tuple(res["pred_future_list"][0].shape) == (V, net["T_p"], 1, 4)
    graph = H.load_diffstg_graph(tmp_path / "g.graphml")
    mean, std = sd["scaler_mean"], sd["scaler_std"]
    torch.manual_seed(12)
    want = []
    for w in range(n_w):
        x = (series[7 * w: 7 * w + net["T_h"]].t().unsqueeze(-1) - mean) / std
        outs = O.evaluation_step(sd, net, x, graph.edge_index, lambda shape: torch.randn(shape))[:, -net["T_p"]:]
        want.append(float((outs * std + mean).var(dim=-1, unbiased=False).mean()))
    print("DiffSTG harness curve", res["ews"], want)
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), np.array(want), rtol=1e-4)
    again = H.uncertainty_ews(model_save_file=tmp_path / "m", data_file=tmp_path / "biomass_x.pt", dynamic_type="biomass",
                              graph_file=tmp_path / "g.graphml", device=dev(), save_nsdiff_g=False, load_model_when_cached=True)
    np.testing.assert_allclose(np.array(again["ews"], dtype=np.float64), np.array(want), rtol=1e-4)        # served from the cache
    bare = H.uncertainty_ews(model_save_file=tmp_path / "m", data_file=tmp_path / "biomass_x.pt", dynamic_type="biomass",
                             graph_file=tmp_path / "g.graphml", device=dev(), save_nsdiff_g=False)
    # as in the reference, a cached call that does not load the model has no scaler: the curve is in scaled units (/ std^2)
    np.testing.assert_allclose(np.array(bare["ews"], dtype=np.float64) * float(std[0]) ** 2, np.array(want), rtol=1e-4)


def test_uncertainty_ews_diffusionts(tmp_path, monkeypatch):
    from conftest import GOLDEN
    from oracle import diffusionts_oracle as O
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    z = np.load(GOLDEN / "diffusionts_cases.npz")
    keys = ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec", "n_heads", "timesteps", "diffusion_steps",
            "n_z_samples", "parallel_sample")
    par = dict(zip(keys, [int(v) for v in z["cfg_small"]]))
    pre = "sd_small_"
    sd = {k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}
    sd["scaler_mean"], sd["scaler_std"] = torch.tensor([0.1, -0.2]), torch.tensor([1.5, 0.7])
    net = dict(par, scaler_type="StandardScaler", noise_source="torch_cpu")
    _write_dir(tmp_path / "m", "DiffusionTS", net, sd, {"windows": par["windows"], "pred_len": par["pred_len"], "sampling_t": 0.1})
    gen = torch.Generator().manual_seed(8)
    series = torch.cumsum(torch.randn(3, 30, 2, generator=gen) * 0.1, dim=1)                      # [Node, T, F]
    _reseed_after(monkeypatch, H, "load_model_from_dir", 13)
    res = H.uncertainty_ews(model_save_file=tmp_path / "m", torch_time_series=series, time_data=torch.arange(30) * 0.1, dynamic_type="SIS",
                            sample_window_step=9, device=dev(), save_nsdiff_g=False, cache_path=tmp_path / "cache")
    n_w = (30 - par["windows"]) // 9 + 1
    assert len(res["ews"]) == n_w and tuple(res["pred_future_list"][0].shape) == (3, par["pred_len"], 2, par["n_z_samples"])
    cfg = O.DTSConfig(par["windows"], par["pred_len"], par["dataset_nf"], par["d_model"], par["n_layer_enc"], par["n_layer_dec"], par["n_heads"],
                      par["timesteps"], par["diffusion_steps"], n_z_samples=par["n_z_samples"], parallel_sample=par["parallel_sample"])
    torch.manual_seed(13)
    want = []
    for w in range(n_w):
        x = (series[:, 9 * w: 9 * w + par["windows"], :] - sd["scaler_mean"]) / sd["scaler_std"]
        outs = O.evaluation_step(sd, cfg, x, lambda shape: torch.randn(shape))
        want.append(float((outs * sd["scaler_std"].view(1, 1, -1, 1) + sd["scaler_mean"].view(1, 1, -1, 1)).var(dim=-1, unbiased=False).mean()))
    print("DiffusionTS harness curve", res["ews"], want)
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), np.array(want), rtol=1e-4)


# ---- sample-axis split (fewer windows than ranks) of the DiffusionTS / DiffSTG / TMDM families: the device side -------------------------
def _union(H, model, windows, pred_len, S, world, graph=None):
    parts = []
    for r in range(world):
        torch.manual_seed(31)
        parts.append(H._native_sample_shard_other(model, windows, pred_len, True, r * (S // world), (r + 1) * (S // world), seed=0, graph=graph).cpu())
    return torch.cat(parts, dim=-1)


def test_sample_axis_shards_union_diffusionts():
    """Rank r samples ALL windows with the chunks [r C/world, (r+1) C/world) of the reference's stream; side by side = the unsplit run."""
    from conftest import GOLDEN
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    z = np.load(GOLDEN / "diffusionts_cases.npz")
    keys = ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec", "n_heads", "timesteps", "diffusion_steps", "n_z_samples", "parallel_sample")
    par = dict(zip(keys, [int(v) for v in z["cfg_small"]]))
    par.update(n_z_samples=8, parallel_sample=2)                                 # 4 chunks of 2
    m = diffusion_models("DiffusionTS", dict(par, device="cpu", scaler_type="StandardScaler", noise_source="torch_cpu"))
    pre = "sd_small_"
    m.load_state_dict({k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}, strict=True)
    m.scaler_mean, m.scaler_std = torch.tensor([0.1, -0.2]), torch.tensor([1.5, 0.7])
    m = m.to(dev()).eval()
    g = torch.Generator().manual_seed(3)
    windows = [torch.cumsum(torch.randn(3, par["windows"], 2, generator=g) * 0.1, dim=1) for _ in range(2)]
    assert H._sample_axis(m) == (8, 2, True)
    torch.manual_seed(31)
    whole, _, _ = H._native_shard(m, windows, par["pred_len"], True)              # [2, 3, O, 2, 8]
    for world in (2, 4):
        got = _union(H, m, windows, par["pred_len"], 8, world)
        assert tuple(got.shape) == tuple(whole.shape) and m.n_z_samples == 8
        assert rel_err(got, whole.cpu()) < 1e-5


def test_sample_axis_shards_union_diffstg():
    from conftest import GOLDEN
    from test_diffstg_oracle import case, state
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    z = np.load(GOLDEN / "diffstg_cases.npz")
    net, V = case(z, "tiny")
    net = dict(net, sequential_sampling=4, noise_source="torch_cpu")             # 4 passes of 2 copies
    m = diffusion_models("DiffSTG", dict(net, device="cpu"))
    m.load_state_dict(state(z, "tiny", net), strict=True)
    m = m.to(dev()).eval()
    graph = H._Graph(T(z["edge_tiny"]), V)
    g = torch.Generator().manual_seed(4)
    windows = [torch.randn(V, net["T_h"], 1, generator=g) * 0.7 for _ in range(2)]
    assert H._sample_axis(m, graph) == (8, 2, True)
    torch.manual_seed(31)
    whole, _, _ = H._native_shard(m, windows, net["T_p"], True, graph=graph)
    for world in (2, 4):
        got = _union(H, m, windows, net["T_p"], 8, world, graph=graph)
        assert tuple(got.shape) == tuple(whole.shape) and m.sequential_sampling == 4
        assert rel_err(got, whole.cpu()) < 1e-5


def test_sample_axis_shards_union_tmdm(golden_tmdm):
    """TMDM samples pointwise: whole chunks (world 2) or a slice of one chunk's parallel_sample axis (world 4, 8)."""
    from test_tmdm import _model
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    m = _model(golden_tmdm, "f2t10", 2, 10, device="cuda:0", noise_source="torch_cpu", n_z_samples=8, parallel_sample=4)
    net = dict(windows=m.windows, pred_len=m.pred_len, dataset_nf=2)
    g = torch.Generator().manual_seed(5)
    windows = [torch.cumsum(torch.randn(3, net["windows"], net["dataset_nf"], generator=g) * 0.1, dim=1) for _ in range(2)]
    assert H._sample_axis(m) == (8, 4, False)
    torch.manual_seed(31)
    whole, _, _ = H._native_shard(m, windows, net["pred_len"], True)
    for world in (2, 4, 8):
        got = _union(H, m, windows, net["pred_len"], 8, world)
        assert tuple(got.shape) == tuple(whole.shape) and (m.n_z_samples, m.parallel_sample) == (8, 4)
        assert rel_err(got, whole.cpu()) < 1e-5
