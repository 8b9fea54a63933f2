"""GPU tests (-m gpu) of Model_spatial, the mean backbone f(x) of NsDiff_spatial (reference NsDiff/mu_backbone.py:186-340, called at
NsDiff_model.py:744), through the C ABI (mpv_nst_load with fT_h > 0, mpv_nst_set_graph, mpv_nst_forward) against
oracle/ns_transformer_oracle.py::forward_spatial, and end to end: NsDiff_model_spatial.evaluation_step and uncertainty_ews WITHOUT a supplied
y_0_hat.

PARITY UNPINNED: the transformer layers are torch-timeseries classes and the graph convolution is torch_geometric's, both absent; the oracle
restates them (its header), the reference's own lines are the wiring, the two k = windows + 1 convolutions and the reshapes around the graph
stage.  Tolerance 1e-4 elementwise."""
import types

import numpy as np
import pytest
import torch
import yaml

from test_gpu_parity import dev, rel_err, _reseed_after

pytestmark = pytest.mark.gpu
RTOL = 1e-4
T = torch.from_numpy

CASES = {
    # tag: (f(x) config, nodes, edges)
    "small": (dict(dataset_nf=2, windows=24, pred_len=12, d_model=32, n_heads=2, d_ff=64, e_layers=1, d_layers=1, fT_h=4, spatial_layers=2), 6,
              [(0, 1), (1, 2), (2, 3), (3, 0), (1, 4)]),                       # node 5 isolated
    "wide": (dict(dataset_nf=1, windows=40, pred_len=20, d_model=64, n_heads=4, d_ff=32, e_layers=2, d_layers=1, activation="relu", fT_h=6,
                  spatial_layers=1), 9, [(i, (i + 1) % 9) for i in range(9)] + [(0, 4), (2, 7)]),
    "no_gnn": (dict(dataset_nf=1, windows=16, pred_len=8, d_model=32, n_heads=2, d_ff=32, e_layers=1, d_layers=1, fT_h=2, spatial_layers=0), 3, [(0, 1)]),
}


def edges(pairs):
    e = torch.tensor(pairs, dtype=torch.int64).t()
    return torch.cat([e, e.flip(0)], dim=1)                                   # both directions, like from_networkx on an undirected graph


def f_only(cfg, sd, **over):
    from mpv_amd.models.models import diffusion_models
    net = dict(device=dev(), scaler_type="StandardScaler", rolling_length=max(2, cfg.seq_len // 4), diffusion_schedule="linear", diffusion_steps=5,
               beta_start=1e-4, beta_end=0.02, n_z_samples=4, parallel_sample=2, windows=cfg.seq_len, pred_len=cfg.pred_len, dataset_nf=cfg.dataset_nf,
               load_pretrain=False, d_model=cfg.d_model, n_heads=cfg.n_heads, d_ff=cfg.d_ff, e_layers=cfg.e_layers, d_layers=cfg.d_layers,
               activation=cfg.activation, p_hidden_dims=list(cfg.p_hidden_dims), p_hidden_layers=cfg.p_hidden_layers, fT_h=cfg.fT_h,
               spatial_layers=cfg.spatial_layers, **over)
    m = diffusion_models("NsDiff_spatial", net, train_model_select="pretrain_f")
    m.cond_pred_model.load_state_dict(sd, strict=True)
    with torch.no_grad():
        m.scaler_mean.zero_()
        m.scaler_std.fill_(1.0)
    return m.to(dev()).eval()


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("tag", list(CASES))
def test_model_spatial_matches_the_restatement(tag, gemm):
    from oracle import ns_transformer_oracle as O
    kw, V, pairs = CASES[tag]
    cfg = O.config(**kw)
    sd = O.random_spatial_state(cfg, seed=len(tag))
    edge = edges(pairs)
    m = f_only(cfg, sd, gemm_mode=gemm)
    assert m.cond_pred_model.gemm_mode == gemm
    g = torch.Generator().manual_seed(2)
    x = torch.randn(V, cfg.seq_len, cfg.dataset_nf, generator=g).cumsum(1) * 0.3
    pred, full = m.cond_pred_model(x.to(dev()), None, edge)
    want_pred, want_full = O.forward_spatial(sd, cfg, x, edge)
    ctx = m.native_context()
    inter = O.intermediates(sd, cfg, x)
    cross, feat = O.spatial_stage(sd, cfg, inter["enc_out"], edge, return_features=True)
    e_cross, e_feat = rel_err(ctx.nst_debug("enc_out", V), cross), rel_err(ctx.nst_debug("spatial", V), feat)
    e_full = rel_err(full, want_full)
    print("Model_spatial %s (%s): graph features %.3e, upsampled encoder memory %.3e, output %.3e" % (tag, gemm, e_feat, e_cross, e_full))
    assert tuple(pred.shape) == (V, cfg.pred_len, cfg.dataset_nf) and torch.equal(full[:, -cfg.pred_len:], pred)
    assert e_feat < RTOL and e_cross < RTOL and e_full < RTOL and rel_err(pred, want_pred) < RTOL


def test_windows_in_one_call_equal_windows_one_by_one_and_whole_windows_per_pass():
    from oracle import ns_transformer_oracle as O
    kw, V, pairs = CASES["small"]
    cfg = O.config(**kw)
    sd = O.random_spatial_state(cfg, seed=9)
    edge = edges(pairs)
    m = f_only(cfg, sd)
    g = torch.Generator().manual_seed(4)
    hist = torch.randn(5, V, cfg.seq_len, cfg.dataset_nf, generator=g).cumsum(2) * 0.3
    both = m.spatial_mean_windows(hist.to(dev()), edge)
    assert tuple(both.shape) == (5, V, cfg.pred_len, cfg.dataset_nf)
    for w in range(5):
        one = m.cond_pred_model(hist[w].to(dev()), None, edge)[0]
        assert rel_err(both[w], one) < 1e-6 and rel_err(both[w], O.forward_spatial(sd, cfg, hist[w], edge)[0]) < RTOL
    # a pass must hold whole windows (the graph couples a window's nodes): 8 rows per pass is rounded down to one 6-node window
    ctx = m.native_context()
    series = hist.permute(1, 0, 2, 3).reshape(V, 5 * cfg.seq_len, cfg.dataset_nf).contiguous().to(dev())
    cut = ctx.nst_forward(series, 5, 0, cfg.seq_len, rows_per_pass=V + 2)
    assert torch.equal(cut, both)


def test_spatial_c_abi_error_paths():
    import ctypes as C
    from mpv_amd import _native
    from oracle import ns_transformer_oracle as O
    kw, V, pairs = CASES["small"]
    cfg = O.config(**kw)
    sd = O.random_spatial_state(cfg, seed=1)
    m = f_only(cfg, sd)
    ctx = m.native_context()
    lib = _native.load_library()
    x = torch.zeros(V, cfg.seq_len, cfg.dataset_nf, device=dev())
    with pytest.raises(_native.MpvError, match="mpv_nst_set_graph"):
        ctx.nst_forward(x, 1)                                                  # no graph yet
    ctx.nst_set_graph(edges(pairs), V)
    with pytest.raises(_native.MpvError, match="mpv_nst_set_graph"):
        ctx.nst_forward(torch.zeros(V + 1, cfg.seq_len, cfg.dataset_nf, device=dev()), 1)      # another node count
    with pytest.raises(_native.MpvError, match="outside"):
        ctx.nst_set_graph(torch.tensor([[0, V], [1, 0]]), V)
    assert bool(torch.isfinite(ctx.nst_forward(x, 1)).all())
    fields, tensors = m.cond_pred_model.native_weight_list()
    d = _native.NstDesc()
    for k, v in fields.items():
        if k == "p_hidden":
            for i, hd in enumerate(v):
                d.p_hidden[i] = int(hd)
        else:
            setattr(d, k, int(v))
    keep = [t.detach().cpu().float().contiguous() for t in tensors]
    arr = (C.POINTER(C.c_float) * len(keep))(*[C.cast(t.data_ptr(), C.POINTER(C.c_float)) for t in keep])
    other = _native.Context(dev())
    d.fT_h = 3                                                                 # odd
    assert lib.mpv_nst_load(other._h, C.byref(d), arr, len(keep)) < 0 and b"fT_h" in lib.mpv_last_error(other._h)
    d.fT_h = fields["fT_h"]
    assert lib.mpv_nst_load(other._h, C.byref(d), arr, len(keep) - 1) < 0     # tensor count
    # a plain Model has no graph stage
    plain_cfg = O.config(**{k: v for k, v in kw.items() if k not in ("fT_h", "spatial_layers")})
    from mpv_amd.models.Diffusion_model.NsDiff import mu_backbone as H
    pm = H.Model(plain_cfg)
    pm.load_state_dict(O.random_state(plain_cfg, 2), strict=True)
    other.load_nst(pm)
    with pytest.raises(_native.MpvError, match="no spatial stage"):
        other.nst_set_graph(edges(pairs), V)
    other.close()


def spatial_net_and_state(z, tag, fx):
    """The reference-run NsDiff_spatial fixture `tag` (sampler + sigma head weights) with a seeded Model_spatial as f(x)."""
    from oracle import ns_transformer_oracle as O
    from test_nsdiff_spatial_oracle import case, state
    net, V = case(z, tag)
    sd = state(z, tag, net)
    cfg = O.config(dataset_nf=net["dataset_nf"], windows=net["windows"], pred_len=net["pred_len"], **fx)
    sd_f = O.random_spatial_state(cfg, seed=17)
    full = dict(sd)
    full.update({"cond_pred_model." + k: v for k, v in sd_f.items()})
    net = dict(net, **fx, activation="gelu", p_hidden_dims=[64, 64], p_hidden_layers=2)
    return net, V, sd, sd_f, full, cfg


FX = dict(d_model=32, n_heads=2, d_ff=32, e_layers=1, d_layers=1, fT_h=4, spatial_layers=1)


def test_evaluation_step_runs_f_g_and_the_sampler_without_a_supplied_mean():
    from conftest import GOLDEN
    from oracle import nsdiff_spatial_oracle as NSS
    from oracle import ns_transformer_oracle as O
    from mpv_amd.models.models import diffusion_models
    z = np.load(GOLDEN / "nsdiff_spatial_cases.npz")
    net, V, sd, sd_f, full, cfg = spatial_net_and_state(z, "tiny", FX)
    m = diffusion_models("NsDiff_spatial", dict(net, device=dev(), noise_source="torch_cpu"), train_model_select="NsDiff_model")
    m.load_state_dict(full, strict=True)
    m = m.to(dev()).eval()
    edge = T(z["edge_tiny"])
    data = types.SimpleNamespace(x=T(z["batch_tiny"]), edge_index=edge, num_nodes=V)
    torch.manual_seed(5)
    outs, truth = m.evaluation_step(data)                                      # no y0hat=: f(x) is the native Model_spatial
    x = T(z["batch_tiny"])[:, : net["windows"], :]
    y0 = O.forward_spatial(sd_f, cfg, x, edge)[0]
    torch.manual_seed(5)
    want = NSS.evaluation_step(sd, net, x, edge, y0, lambda shape: torch.randn(shape))
    err = rel_err(outs, want)
    mpv_g, mpv_w = outs.var(dim=-1, unbiased=False).mean().item(), want.var(dim=-1, unbiased=False).mean().item()
    print("NsDiff_spatial evaluation_step with native f(x): rel err %.3e, MPV %.6g vs %.6g" % (err, mpv_g, mpv_w))
    assert truth is None and err < RTOL and abs(mpv_g - mpv_w) <= RTOL * abs(mpv_w)


def test_uncertainty_ews_on_a_nsdiff_spatial_directory(tmp_path, monkeypatch):
    """uncertainty_ews end to end on a NsDiff_spatial model directory (reference layout: model_trained + model_trained.yaml), a data record and a
    graphml topology -- f(x), g(x), sampler and MPV per window on the device -- against the oracles driven window by window on the same torch seed."""
    import networkx as nx
    from conftest import GOLDEN
    from oracle import nsdiff_spatial_oracle as NSS
    from oracle import ns_transformer_oracle as O
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    from mpv_amd.utils.utils import save_checkpoint
    z = np.load(GOLDEN / "nsdiff_spatial_cases.npz")
    net, V, sd, sd_f, full, cfg = spatial_net_and_state(z, "tiny", FX)
    full["scaler_mean"], full["scaler_std"] = torch.tensor([0.8] * net["dataset_nf"]), torch.tensor([1.7] * net["dataset_nf"])
    path = tmp_path / "m"
    path.mkdir()
    mm = diffusion_models("NsDiff_spatial", dict(net, device="cpu"), train_model_select="NsDiff_model")
    mm.load_state_dict(full, strict=True)
    save_checkpoint(str(path), "model_trained", mm, dict(net, task_model="NsDiff_spatial", device="cpu"))
    step = 5
    (path / "model_trained.yaml").write_text(yaml.safe_dump({
        "dataset": {"windows": net["windows"], "pred_len": net["pred_len"], "sampling_t": 0.1},
        "net": dict(net, task_model="NsDiff_spatial", device="cpu"), "train": {"train_model_select": "NsDiff_model"}}))
    edge = z["edge_tiny"]
    g = nx.Graph()
    g.add_nodes_from(range(V))
    g.add_edges_from([(int(a), int(b)) for a, b in edge.T if a < b])
    nx.write_graphml(g, tmp_path / "g.graphml")
    ticks = net["windows"] + 3 * step
    gen = torch.Generator().manual_seed(3)
    series = torch.cumsum(torch.randn(ticks, V, generator=gen) * 0.2, dim=0) + 1.0                 # [ticks, V]
    torch.save({"ys_dynamic": series, "ts_dynamic": torch.arange(ticks) * 0.1, "tp_values": torch.zeros(1)}, tmp_path / "biomass_x.pt")
    _reseed_after(monkeypatch, H, "load_model_from_dir", 21)
    res = H.uncertainty_ews(model_save_file=path, data_file=tmp_path / "biomass_x.pt", dynamic_type="biomass", graph_file=tmp_path / "g.graphml",
                            device=dev(), save_nsdiff_g=False, sample_window_step=step, infer_params={"noise_source": "torch_cpu"})
    n_w = (ticks - net["windows"]) // step + 1
    assert len(res["ews"]) == n_w and tuple(res["pred_future_list"][0].shape) == (V, net["pred_len"], net["dataset_nf"], net["n_z_samples"])
    graph = H.load_diffstg_graph(tmp_path / "g.graphml")
    mean, std = full["scaler_mean"], full["scaler_std"]
    torch.manual_seed(21)
    want = []
    for w in range(n_w):
        x = (series[step * w: step * w + net["windows"]].t().unsqueeze(-1).expand(-1, -1, net["dataset_nf"]) - mean) / std
        y0 = O.forward_spatial(sd_f, cfg, x, graph.edge_index)[0]
        outs = NSS.evaluation_step(sd, net, x, graph.edge_index, y0, lambda shape: torch.randn(shape))
        want.append(float((outs * std.view(1, 1, -1, 1) + mean.view(1, 1, -1, 1)).var(dim=-1, unbiased=False).mean()))
    print("NsDiff_spatial harness curve", res["ews"], want)
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), np.array(want), rtol=1e-4)
    again = H.uncertainty_ews(model_save_file=path, data_file=tmp_path / "biomass_x.pt", dynamic_type="biomass", graph_file=tmp_path / "g.graphml",
                              device=dev(), save_nsdiff_g=False, sample_window_step=step, load_model_when_cached=True)
    np.testing.assert_allclose(np.array(again["ews"], dtype=np.float64), np.array(want), rtol=1e-4)        # served from the cache


def test_sample_axis_shards_union_nsdiff_spatial():
    """Fewer windows than ranks: rank r samples ALL windows (f(x), g(x) everywhere) with the chunks [r C/world, (r+1) C/world) of the reference's stream;
    side by side = the unsplit run (the device side of distributed._sample_split for this family)."""
    from conftest import GOLDEN
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    z = np.load(GOLDEN / "nsdiff_spatial_cases.npz")
    net, V, sd, sd_f, full, cfg = spatial_net_and_state(z, "tiny", FX)
    net = dict(net, n_z_samples=8, parallel_sample=2, noise_source="torch_cpu")            # 4 chunks of 2 graph copies
    m = diffusion_models("NsDiff_spatial", dict(net, device="cpu"), train_model_select="NsDiff_model")
    m.load_state_dict(full, strict=True)
    with torch.no_grad():
        m.scaler_mean.fill_(0.1)
        m.scaler_std.fill_(1.3)
    m = m.to(dev()).eval()
    graph = H._Graph(T(z["edge_tiny"]), V)
    g = torch.Generator().manual_seed(4)
    windows = [torch.cumsum(torch.randn(V, net["windows"], 1, generator=g) * 0.3, dim=1) for _ in range(2)]
    assert H._sample_axis(m, graph) == (8, 2, True)
    torch.manual_seed(31)
    whole, _, _ = H._native_shard(m, windows, net["pred_len"], True, graph=graph)           # [2, V, O, 1, 8]
    for world in (2, 4):
        parts = []
        for r in range(world):
            torch.manual_seed(31)
            parts.append(H._native_sample_shard_other(m, windows, net["pred_len"], True, r * (8 // world), (r + 1) * (8 // world), seed=0, graph=graph).cpu())
        got = torch.cat(parts, dim=-1)
        assert tuple(got.shape) == tuple(whole.shape) and int(m.configs.n_z_samples) == 8
        assert rel_err(got, whole.cpu()) < 1e-5
