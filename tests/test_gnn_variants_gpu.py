"""GPU parity (-m gpu) of the graph-convolution variants of a SpatialBlock (SAGEConv mean / add, GraphConv, GCNConv, GATConv; reference
models/layer/gnn_conv.py:4-28) through the C ABI, in DiffSTG's and NsDiff_spatial's UGnet, against the CPU oracle's restatements (UNPINNED:
torch_geometric is absent, no reference fixture contains these layers).  Tolerance 1e-4, elementwise, as for the gated layer."""
import types

import numpy as np
import pytest
import torch

from oracle import diffstg_oracle as O
from test_gnn_variants import VARIANTS
from test_gpu_parity import dev, rel_err

pytestmark = pytest.mark.gpu
T = torch.from_numpy
RTOL = 1e-4


def seeded(model, seed):
    torch.manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if ".spatial.gnn." in name:
                p.copy_(torch.randn_like(p) * (0.5 if p.dim() > 1 else 0.2))          # biases and attention vectors away from their zero / xavier init
    return model


def with_self_loop(edge):
    """The fixture's graph + one self loop and one parallel edge: GCN / GAT replace input self loops by one per node, sums count parallel edges."""
    extra = torch.tensor([[2, int(edge[0, 0])], [2, int(edge[1, 0])]], dtype=edge.dtype)
    return torch.cat((edge, extra), dim=1)


@pytest.mark.parametrize("gemm", ["f16x3", "f32"])
@pytest.mark.parametrize("name,param", VARIANTS)
def test_diffstg_ugnet_with_the_variant_matches_the_oracle(name, param, gemm):
    from conftest import GOLDEN
    from mpv_amd.models.models import diffusion_models
    from test_diffstg_oracle import case
    z = np.load(GOLDEN / "diffstg_cases.npz")
    for tag in ("tiny", "ddpm"):
        net, V = case(z, tag)
        net = dict(net, gnn_name=name, gnn_param=param)
        torch.manual_seed(5)
        m = seeded(diffusion_models("DiffSTG", dict(net, device=dev(), gemm_mode=gemm)), 7).to(dev()).eval()
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        edge = with_self_loop(T(z["edge_%s" % tag]))
        ctx = m.native_context()
        ctx.stg_set_graph(edge, V)
        x, xm = T(z["x_%s" % tag]), T(z["xm_%s" % tag])
        for t in (1, net["diffusion_steps"]):
            want = O.ugnet(sd, net, x, torch.tensor([float(t)]), xm, O.duplicate_edge_index(x.shape[0] // V, edge, V))
            err = rel_err(ctx.stg_forward(x.to(dev()), xm.to(dev()), t), want)
            print("DiffSTG UGnet %s %s %s t=%d (%s) rel err %.3e" % (name, param.get("aggr", ""), tag, t, gemm, err))
            assert err < RTOL


@pytest.mark.parametrize("name,param", [VARIANTS[0], VARIANTS[3], VARIANTS[4]])
def test_diffstg_evaluation_step_with_the_variant_matches_the_oracle(name, param):
    """The whole sampler (graph copies, DDIM / DDPM plan, the oracle's draws replayed in order) on a SAGEConv / GCNConv / GATConv network."""
    from conftest import GOLDEN
    from mpv_amd.models.models import diffusion_models
    from test_diffstg_oracle import case
    z = np.load(GOLDEN / "diffstg_cases.npz")
    net, V = case(z, "tiny")
    net = dict(net, gnn_name=name, gnn_param=param)
    torch.manual_seed(5)
    m = seeded(diffusion_models("DiffSTG", dict(net, device=dev())), 9).to(dev()).eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    batch, edge = T(z["batch_tiny"]), T(z["edge_tiny"])
    g, draws = torch.Generator().manual_seed(31), []

    def noise(shape):
        draws.append(torch.randn(shape, generator=g))
        return draws[-1]

    want = O.evaluation_step(sd, net, batch, edge, noise)
    data = types.SimpleNamespace(x=batch, edge_index=edge, num_nodes=V)
    outs, _ = m.evaluation_step(data, noise=torch.cat([d.reshape(-1) for d in draws]))
    err = rel_err(outs, want)
    print("DiffSTG evaluation_step %s rel err %.3e" % (name, err))
    assert tuple(outs.shape) == tuple(want.shape) and err < RTOL


@pytest.mark.parametrize("name,param", VARIANTS)
def test_nsdiff_spatial_ugnet_with_the_variant_matches_the_oracle(name, param):
    from conftest import GOLDEN
    from oracle import nsdiff_spatial_oracle as NSS
    from mpv_amd.models.models import diffusion_models
    from test_nsdiff_spatial_oracle import case
    z = np.load(GOLDEN / "nsdiff_spatial_cases.npz")
    net, V = case(z, "f2")
    net = dict(net, gnn_name=name, gnn_param=param)
    torch.manual_seed(5)
    m = seeded(diffusion_models("NsDiff_spatial", dict(net, device=dev()), train_model_select="NsDiff_model"), 13).to(dev()).eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    edge = with_self_loop(T(z["edge_f2"]))
    ctx = m.native_context()
    ctx.nss_set_graph(edge, V)
    y, y0, gx = (T(z["%s_f2" % k]) for k in ("y", "y0", "gx"))
    pe = NSS.duplicate_edge_index(y.shape[0] // V, edge, V)
    for t in (0, net["diffusion_steps"] - 1):
        eps, sig = ctx.nss_forward(y.to(dev()), y0.to(dev()), gx.to(dev()), t)
        we, ws = NSS.ugnet(sd, net, y, y0, gx, torch.tensor([t]), pe)
        e1, e2 = rel_err(eps, we), rel_err(sig, ws)
        print("NsDiff_spatial UGnet %s t=%d: eps rel err %.3e, sigma rel err %.3e" % (name, t, e1, e2))
        assert e1 < RTOL and e2 < RTOL


@pytest.mark.parametrize("name,param", VARIANTS)
def test_model_spatial_with_the_variant_matches_the_oracle(name, param):
    """f(x) of NsDiff_spatial with f_gnn_name / f_gnn_param (reference NsDiff/mu_backbone.py:224): the graph stage of csrc/nst.hip on the same
    product + gather as the UGnet's blocks (csrc/graph_aggr.h)."""
    from oracle import ns_transformer_oracle as NO
    from test_gnn_variants import holder_state
    from test_nst_spatial_gpu import CASES, edges, f_only
    kw, V, pairs = CASES["small"]
    cfg = NO.config(**kw)
    cfg.f_gnn_name, cfg.f_gnn_param = name, param
    sd = {k: v for k, v in NO.random_spatial_state(cfg, seed=21).items() if ".gnn." not in k}
    C = cfg.fT_h * cfg.d_model
    for i in range(cfg.spatial_layers):
        _, layer = holder_state(name, param, C, 40 + i)
        sd.update({"spatial_encoder.%d.gnn.%s" % (i, k[2:]): v * (0.3 if v.dim() > 1 else 1.0) for k, v in layer.items()})
    edge = with_self_loop(edges(pairs))
    m = f_only(cfg, sd, f_gnn_name=name, f_gnn_param=param)
    x = torch.randn(V, cfg.seq_len, cfg.dataset_nf, generator=torch.Generator().manual_seed(2)).cumsum(1) * 0.3
    pred, full = m.cond_pred_model(x.to(dev()), None, edge)
    want_pred, want_full = NO.forward_spatial(sd, cfg, x, edge)
    feat = NO.spatial_stage(sd, cfg, NO.intermediates(sd, cfg, x)["enc_out"], edge, return_features=True)[1]
    e_feat, e_full = rel_err(m.native_context().nst_debug("spatial", V), feat), rel_err(full, want_full)
    print("Model_spatial %s: graph features %.3e, output %.3e" % (name, e_feat, e_full))
    assert e_feat < RTOL and e_full < RTOL and rel_err(pred, want_pred) < RTOL


@pytest.mark.parametrize("name,param", [VARIANTS[0], VARIANTS[4]])
def test_uncertainty_ews_on_a_model_directory_with_the_variant(name, param, tmp_path, monkeypatch):
    """The harness end to end on a DiffSTG model directory whose YAML names SAGEConv / GATConv (gnn_param through the YAML round trip), against
    the oracle driven window by window on the same draws."""
    import networkx as nx
    from conftest import GOLDEN
    from mpv_amd.evaluation_and_analysis import diffusion_model_uncertainy as H
    from mpv_amd.models.models import diffusion_models
    from test_diffstg_oracle import case
    from test_gpu_parity import _reseed_after
    from test_harness_families_gpu import _write_dir
    z = np.load(GOLDEN / "diffstg_cases.npz")
    net, V = case(z, "tiny")
    net = dict(net, gnn_name=name, gnn_param=param, noise_source="torch_cpu")
    torch.manual_seed(5)
    m = seeded(diffusion_models("DiffSTG", dict(net, device="cpu")), 17)
    with torch.no_grad():
        m.scaler_mean.fill_(0.9)
        m.scaler_std.fill_(0.7)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    _write_dir(tmp_path / "m", "DiffSTG", net, sd, {"windows": net["T_h"], "pred_len": net["T_p"], "sampling_t": 0.1, "interval_step": 9})
    g = nx.Graph()
    g.add_nodes_from(range(V))
    g.add_edges_from([(int(a), int(b)) for a, b in z["edge_tiny"].T if a < b])
    nx.write_graphml(g, tmp_path / "g.graphml")
    series = torch.cumsum(torch.randn(40, V, generator=torch.Generator().manual_seed(3)) * 0.2, dim=0) + 1.0
    torch.save({"ys_dynamic": series, "ts_dynamic": torch.arange(40) * 0.1, "tp_values": torch.zeros(1)}, tmp_path / "biomass_x.pt")
    _reseed_after(monkeypatch, H, "load_model_from_dir", 12)
    res = H.uncertainty_ews(model_save_file=tmp_path / "m", data_file=tmp_path / "biomass_x.pt", dynamic_type="biomass",
                            graph_file=tmp_path / "g.graphml", device=dev(), save_nsdiff_g=False, infer_params={"noise_source": "torch_cpu"})
    graph = H.load_diffstg_graph(tmp_path / "g.graphml")
    mean, std = sd["scaler_mean"], sd["scaler_std"]
    torch.manual_seed(12)
    want = []
    for w in range((40 - net["T_h"]) // 9 + 1):
        x = (series[9 * w: 9 * w + net["T_h"]].t().unsqueeze(-1) - mean) / std
        outs = O.evaluation_step(sd, net, x, graph.edge_index, lambda shape: torch.randn(shape))[:, -net["T_p"]:]
        want.append(float((outs * std + mean).var(dim=-1, unbiased=False).mean()))
    print("DiffSTG (%s) harness curve" % name, res["ews"], want)
    np.testing.assert_allclose(np.array(res["ews"], dtype=np.float64), np.array(want), rtol=1e-4)
