"""Weight holders of the graph convolutions `gnn_conv` selects (mirrors reference models/layer/gnn_conv.py:4-28 in the factory's signature and in
torch_geometric's parameter NAMES and shapes, so a reference checkpoint loads strictly).  Nothing here computes: a SpatialBlock's layer runs inside
the HIP engines (csrc/stg.hip: one product for the layer's Linear maps + one gather over the incoming edges), which `native_kind()` /
`native_tensors()` describe to the loader (include/mpv_hip.h, mpv_stg_step kinds).  torch_geometric is absent from this image: every layer is
restated from its published definition (parity unpinned, like ResGatedGraphConv).

Native: ResGatedGraphConv, SAGEConv (aggr mean | add | sum), GraphConv (aggr add | sum | mean), GCNConv, GATConv (concat=True).
Not native (NotImplementedError, as before): GATv2Conv, GatedGraphConv, PNAConv, and the reference's own AGNNConv / FGNNConv."""
from __future__ import annotations

import torch
import torch.nn as nn

KIND_GATED, KIND_SAGE_MEAN, KIND_SUM, KIND_GCN, KIND_GAT = 0, 10, 11, 12, 13          # include/mpv_hip.h: mpv_stg_step.kind of a residual block


def _only(name, kwargs, allowed):
    extra = sorted(set(kwargs) - set(allowed))
    if extra:
        raise ValueError("unsupported {} options: {}".format(name, extra))


class ResGatedGraphConv(nn.Module):
    def __init__(self, in_channels, out_channels, root_weight=True, bias=True, **kwargs):
        super().__init__()
        if kwargs:
            raise ValueError("unsupported ResGatedGraphConv options: {}".format(sorted(kwargs)))
        self.lin_key = nn.Linear(in_channels, out_channels)
        self.lin_query = nn.Linear(in_channels, out_channels)
        self.lin_value = nn.Linear(in_channels, out_channels)
        self.lin_skip = nn.Linear(in_channels, out_channels, bias=False) if root_weight else None
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def native_kind(self):
        return KIND_GATED

    def native_tensors(self):
        """w[12..19] of the step record."""
        return [self.lin_key.weight, self.lin_key.bias, self.lin_query.weight, self.lin_query.bias, self.lin_value.weight, self.lin_value.bias,
                self.lin_skip.weight if self.lin_skip is not None else torch.zeros_like(self.lin_key.weight),
                self.bias if self.bias is not None else torch.zeros_like(self.lin_key.bias)]


class SAGEConv(nn.Module):
    """torch_geometric.nn.SAGEConv(in, out, aggr="mean", normalize=False, root_weight=True, project=False, bias=True):
    out_i = lin_l(aggr_{j->i} x_j) + lin_r(x_i); lin_l carries the bias, lin_r has none."""

    def __init__(self, in_channels, out_channels, aggr="mean", root_weight=True, bias=True, normalize=False, project=False, **kwargs):
        super().__init__()
        _only("SAGEConv", kwargs, ())
        if normalize or project:
            raise NotImplementedError("SAGEConv(normalize=True / project=True) is not part of the native path")
        if aggr not in ("mean", "add", "sum"):
            raise NotImplementedError("SAGEConv aggr={!r}: the native path has mean and add".format(aggr))
        self.aggr = aggr
        self.lin_l = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_r = nn.Linear(in_channels, out_channels, bias=False) if root_weight else None

    def native_kind(self):
        return KIND_SAGE_MEAN if self.aggr == "mean" else KIND_SUM

    def native_tensors(self):
        return [self.lin_l.weight, self.lin_l.bias, self.lin_r.weight if self.lin_r is not None else None, None, None, None, None, None]


class GraphConv(nn.Module):
    """torch_geometric.nn.GraphConv(in, out, aggr="add", bias=True): out_i = lin_rel(aggr_{j->i} x_j) + lin_root(x_i) (unit edge weights)."""

    def __init__(self, in_channels, out_channels, aggr="add", bias=True, **kwargs):
        super().__init__()
        _only("GraphConv", kwargs, ())
        if aggr not in ("mean", "add", "sum"):
            raise NotImplementedError("GraphConv aggr={!r}: the native path has add and mean".format(aggr))
        self.aggr = aggr
        self.lin_rel = nn.Linear(in_channels, out_channels, bias=bias)
        self.lin_root = nn.Linear(in_channels, out_channels, bias=False)

    def native_kind(self):
        return KIND_SAGE_MEAN if self.aggr == "mean" else KIND_SUM

    def native_tensors(self):
        return [self.lin_rel.weight, self.lin_rel.bias, self.lin_root.weight, None, None, None, None, None]


class GCNConv(nn.Module):
    """torch_geometric.nn.GCNConv(in, out, improved=False, add_self_loops=True, normalize=True, bias=True):
    out = D^-1/2 (A + I) D^-1/2 (x W^T) + bias, D = in-degree + 1."""

    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True, normalize=True, bias=True, **kwargs):
        super().__init__()
        _only("GCNConv", kwargs, ())
        if improved or not add_self_loops or not normalize:
            raise NotImplementedError("GCNConv: the native path has the default improved=False, add_self_loops=True, normalize=True")
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None

    def native_kind(self):
        return KIND_GCN

    def native_tensors(self):
        return [self.lin.weight, self.bias, None, None, None, None, None, None]


class GATConv(nn.Module):
    """torch_geometric.nn.GATConv(in, out_per_head, heads, concat=True, negative_slope=0.2, dropout, add_self_loops=True, bias=True) in eval mode
    (the attention dropout is inactive): x' = lin(x) viewed [H, D]; e_ij = leaky_relu(<x'_j, att_src> + <x'_i, att_dst>);
    out_i = concat_h sum_j softmax_j(e_ij) x'_j + bias.  Releases that keep lin_src / lin_dst (one shared weight) load into `lin`."""

    def __init__(self, in_channels, out_channels, heads=1, concat=True, negative_slope=0.2, dropout=0.0, add_self_loops=True, bias=True, **kwargs):
        super().__init__()
        _only("GATConv", kwargs, ())
        if not concat or not add_self_loops:
            raise NotImplementedError("GATConv: the native path has concat=True, add_self_loops=True")
        self.heads, self.out_channels, self.negative_slope, self.dropout = int(heads), int(out_channels), float(negative_slope), float(dropout)
        self.lin = nn.Linear(in_channels, self.heads * self.out_channels, bias=False)
        self.att_src = nn.Parameter(torch.empty(1, self.heads, self.out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, self.heads, self.out_channels))
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)
        self.bias = nn.Parameter(torch.zeros(self.heads * self.out_channels)) if bias else None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for old in ("lin_src.weight", "lin_dst.weight"):                     # torch_geometric 2.0-2.4: the same tensor under two names
            if prefix + old in state_dict:
                w = state_dict.pop(prefix + old)
                state_dict.setdefault(prefix + "lin.weight", w)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def native_kind(self):
        return KIND_GAT

    def native_tensors(self):
        meta = torch.tensor([float(self.heads), self.negative_slope], dtype=torch.float32)
        return [self.lin.weight, self.bias, self.att_src.reshape(self.heads, self.out_channels), self.att_dst.reshape(self.heads, self.out_channels),
                meta, None, None, None]


_NOT_NATIVE = ("GATv2Conv", "GatedGraphConv", "PNAConv", "AGNNConv", "FGNNConv")


def gnn_conv(gnn_name, in_channels, out_channels, gnn_param: dict = None):
    """The reference's factory (models/layer/gnn_conv.py:4-28): same names, same treatment of `heads` for GATConv."""
    gnn_param = dict(gnn_param or {})
    if gnn_name == "ResGatedGraphConv":
        return ResGatedGraphConv(in_channels, out_channels, **gnn_param)
    if gnn_name == "SAGEConv":
        return SAGEConv(in_channels, out_channels, **gnn_param)
    if gnn_name == "GraphConv":
        return GraphConv(in_channels, out_channels, **gnn_param)
    if gnn_name == "GCNConv":
        return GCNConv(in_channels, out_channels, **gnn_param)
    if gnn_name == "GATConv":
        if "heads" not in gnn_param:
            raise KeyError("heads")                                          # the reference indexes gnn_param["heads"] (gnn_conv.py:7)
        if out_channels % int(gnn_param["heads"]) != 0:
            raise ValueError("GATConv: heads = {} must divide the {} output channels".format(gnn_param["heads"], out_channels))
        return GATConv(in_channels, out_channels // int(gnn_param["heads"]), **gnn_param)
    if gnn_name in _NOT_NATIVE:
        raise NotImplementedError("gnn_name={!r} has no native kernel (native: ResGatedGraphConv, SAGEConv, GraphConv, GCNConv, GATConv)".format(gnn_name))
    raise ValueError("the definition  don't exit\n\tyou can define it before using it")
