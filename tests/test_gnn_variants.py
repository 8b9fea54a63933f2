"""CPU tests of the graph convolutions `gnn_conv` can put into a SpatialBlock besides ResGatedGraphConv (reference models/layer/gnn_conv.py:4-28;
configs/grid_search/diffusion_model_DiffSTG.yaml:39 lists GATConv and SAGEConv): the oracle's restatements against the layers' dense-matrix
definitions, and the parameter holders' names / factory behaviour.  torch_geometric is absent: parity of these layers is UNPINNED."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import diffstg_oracle as O

VARIANTS = [("SAGEConv", {"aggr": "mean"}), ("SAGEConv", {"aggr": "add"}), ("GraphConv", {}), ("GCNConv", {}),
            ("GATConv", {"heads": 2, "concat": True, "dropout": 0.05, "negative_slope": 0.2})]


def random_graph(n, e, seed, self_loops=False):
    g = torch.Generator().manual_seed(seed)
    src, dst = torch.randint(0, n, (e,), generator=g), torch.randint(0, n, (e,), generator=g)
    src[0], dst[0] = 1, 0
    dst[dst == n - 1] = 0                                   # node n - 1 has no incoming edge: the empty aggregation
    if not self_loops:
        keep = src != dst
        src, dst = src[keep], dst[keep]
    return torch.stack((src, dst))


def holder_state(name, param, c, seed):
    from mpv_amd.models.layer.gnn_conv import gnn_conv
    torch.manual_seed(seed)
    m = gnn_conv(name, c, c, param)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    return m, {"g." + k: v.detach().clone() for k, v in m.state_dict().items()}


def adjacency(edge, n):
    a = torch.zeros(n, n, dtype=torch.float64)
    for s, d in edge.t().tolist():
        a[d, s] += 1.0                                      # a[i][j]: edges j -> i
    return a


@pytest.mark.parametrize("name,param", VARIANTS)
def test_restatement_equals_the_dense_definition(name, param):
    n, c = 9, 8
    edge = random_graph(n, 30, 5)
    _, sd = holder_state(name, param, c, 11)
    x = torch.randn(n, c, generator=torch.Generator().manual_seed(3))
    got = O.graph_conv(sd, "g.", x, edge, param).double()
    a, xd = adjacency(edge, n), x.double()
    w = {k: v.double() for k, v in sd.items()}
    if name in ("SAGEConv", "GraphConv"):
        nbr, root = ("g.lin_l.", "g.lin_r.") if name == "SAGEConv" else ("g.lin_rel.", "g.lin_root.")
        agg = a @ xd
        if param.get("aggr", "mean" if name == "SAGEConv" else "add") == "mean":
            agg = agg / a.sum(1).clamp(min=1.0)[:, None]
        want = agg @ w[nbr + "weight"].t() + w[nbr + "bias"] + xd @ w[root + "weight"].t()
        assert float(agg[n - 1].abs().sum()) == 0.0
    elif name == "GCNConv":
        ah = a - torch.diag(torch.diag(a)) + torch.eye(n, dtype=torch.float64)         # one self loop per node, whatever the input had
        dis = ah.sum(1).pow(-0.5)
        want = (dis[:, None] * ah * dis[None, :]) @ (xd @ w["g.lin.weight"].t()) + w["g.bias"]
    else:
        H = param["heads"]
        h = (xd @ w["g.lin.weight"].t()).view(n, H, -1)
        s_src, s_dst = (h * w["g.att_src"].view(1, H, -1)).sum(-1), (h * w["g.att_dst"].view(1, H, -1)).sum(-1)
        ah = a - torch.diag(torch.diag(a)) + torch.eye(n, dtype=torch.float64)         # multiplicity of edge j -> i (parallel edges count twice)
        want = torch.zeros(n, H, c // H, dtype=torch.float64)
        for i in range(n):
            e = F.leaky_relu(s_src + s_dst[i][None, :], param["negative_slope"])           # [j, H]
            wgt = ah[i][:, None] * torch.exp(e - e[ah[i] > 0].max(0).values[None, :])
            want[i] = ((wgt / wgt.sum(0, keepdim=True))[:, :, None] * h).sum(0)
        want = want.reshape(n, c) + w["g.bias"]
    assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())


def test_gat_drops_input_self_loops_and_adds_one():
    n, c, param = 6, 4, {"heads": 2}
    _, sd = holder_state("GATConv", param, c, 2)
    x = torch.randn(n, c, generator=torch.Generator().manual_seed(8))
    edge = random_graph(n, 14, 9)
    with_loops = torch.cat((edge, torch.tensor([[2, 2, 4], [2, 2, 4]])), dim=1)
    assert torch.allclose(O.graph_conv(sd, "g.", x, edge, param), O.graph_conv(sd, "g.", x, with_loops, param), atol=1e-7)


def test_holders_carry_torch_geometric_names_and_the_factory_mirrors_the_reference():
    from mpv_amd.models.layer import gnn_conv as G
    keys = lambda m: sorted(m.state_dict())
    assert keys(G.gnn_conv("SAGEConv", 8, 8, {"aggr": "mean"})) == ["lin_l.bias", "lin_l.weight", "lin_r.weight"]
    assert keys(G.gnn_conv("GraphConv", 8, 8, {})) == ["lin_rel.bias", "lin_rel.weight", "lin_root.weight"]
    assert keys(G.gnn_conv("GCNConv", 8, 8, {})) == ["bias", "lin.weight"]
    gat = G.gnn_conv("GATConv", 8, 8, {"heads": 2, "concat": True, "dropout": 0.05, "negative_slope": 0.2})
    assert keys(gat) == ["att_dst", "att_src", "bias", "lin.weight"] and tuple(gat.att_src.shape) == (1, 2, 4) and tuple(gat.lin.weight.shape) == (8, 8)
    old = {"att_src": gat.att_src.detach(), "att_dst": gat.att_dst.detach(), "bias": gat.bias.detach(), "lin_src.weight": torch.ones(8, 8),
           "lin_dst.weight": torch.ones(8, 8)}
    gat.load_state_dict(old, strict=True)                                      # torch_geometric 2.0-2.4 key names
    assert float(gat.lin.weight.sum()) == 64.0
    assert G.gnn_conv("SAGEConv", 8, 8, {"aggr": "add"}).native_kind() == G.KIND_SUM and G.gnn_conv("GraphConv", 8, 8, {}).native_kind() == G.KIND_SUM
    assert G.gnn_conv("GraphConv", 8, 8, {"aggr": "mean"}).native_kind() == G.KIND_SAGE_MEAN
    with pytest.raises(KeyError):
        G.gnn_conv("GATConv", 8, 8, {})                                        # the reference indexes gnn_param["heads"]
    for name in ("GATv2Conv", "GatedGraphConv", "PNAConv", "AGNNConv", "FGNNConv"):
        with pytest.raises(NotImplementedError):
            G.gnn_conv(name, 8, 8, {})
    with pytest.raises(ValueError):
        G.gnn_conv("NoSuchConv", 8, 8, {})
    with pytest.raises(NotImplementedError):
        G.gnn_conv("GATConv", 8, 8, {"heads": 2, "concat": False})
    import re
    from pathlib import Path
    header = (Path(__file__).resolve().parents[1] / "include" / "mpv_hip.h").read_text()
    for macro, kind in (("MPV_STG_RES_SAGE_MEAN", G.KIND_SAGE_MEAN), ("MPV_STG_RES_SUM", G.KIND_SUM), ("MPV_STG_RES_GCN", G.KIND_GCN), ("MPV_STG_RES_GAT", G.KIND_GAT)):
        assert int(re.search(r"#define %s (\d+)" % macro, header).group(1)) == kind


@pytest.mark.parametrize("name,param", VARIANTS)
def test_models_build_with_the_variant_and_the_oracle_runs_them(name, param):
    """DiffSTG and NsDiff_spatial construct with every native gnn_name (they raised NotImplementedError before round 4), their state dicts carry
    the layer's names, and the oracle's UGnet runs on them."""
    from mpv_amd.models.models import diffusion_models
    from test_diffstg_oracle import case
    from conftest import GOLDEN
    z = np.load(GOLDEN / "diffstg_cases.npz")
    net, V = case(z, "tiny")
    net = dict(net, gnn_name=name, gnn_param=param)
    torch.manual_seed(3)
    m = diffusion_models("DiffSTG", dict(net))
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    first = {"SAGEConv": "lin_l.weight", "GraphConv": "lin_rel.weight", "GCNConv": "lin.weight", "GATConv": "att_src"}[name]
    assert any(k.endswith("spatial.gnn." + first) for k in sd)
    x, xm, edge = torch.from_numpy(z["x_tiny"]), torch.from_numpy(z["xm_tiny"]), torch.from_numpy(z["edge_tiny"])
    eps = O.ugnet(sd, net, x, torch.tensor([3.0]), xm, edge)
    assert tuple(eps.shape) == tuple(x.shape) and bool(torch.isfinite(eps).all())
