"""CPU oracle for the DiffSTG sampling path.

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() as the checker; the
product (package mpv_amd) never imports it.

Parity status
  * PINNED against the reference's own code (tests/golden/make_golden_diffstg.py): GaussianDiffusion tables, InferenceSchedule,
    gaussian_posterior, and the whole of UGnet.forward / DiffSTG.evaluation_step EXCEPT the graph convolution -- the reference's
    modules are run with their `spatial.gnn` (torch_geometric.nn.ResGatedGraphConv, absent here) replaced by the restatement
    below, so TCN blocks, LayerNorm, the k = T+1 down / up-sampling convolutions, time embedding, U-Net wiring, sampler and
    tiling are the reference's arithmetic.
  * UNPINNED: `res_gated_graph_conv` restates torch_geometric 2.5.3's ResGatedGraphConv from its published definition
    (out_i = W_skip x_i + sum_{j -> i} sigmoid(W_k x_i + b_k + W_q x_j + b_q) * (W_v x_j + b_v) + bias); no run of the real
    layer is available (package absent, no network).  Likewise `sage_conv` (SAGEConv / GraphConv), `gcn_conv`, `gat_conv` (round 4: the
    other layers models/layer/gnn_conv.py:4-28 can select; no reference fixture contains them).

Paths are relative to models/Diffusion_model/DiffSTG/ in the reference checkout.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

NoiseFn = Callable[[Tuple[int, ...]], torch.Tensor]
NET = "model."


def time_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:                      # ugnet.py:15-33
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half, dtype=torch.float32) * -emb)
    emb = t.float()[:, None] * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def res_gated_graph_conv(sd, p: str, x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """torch_geometric.nn.ResGatedGraphConv(in, out, root_weight=True, bias=True), aggr='add', flow source_to_target:
    edge_index[0] = source j, edge_index[1] = target i.  UNPINNED restatement (see module docstring)."""
    k = F.linear(x, sd[p + "lin_key.weight"], sd[p + "lin_key.bias"])
    q = F.linear(x, sd[p + "lin_query.weight"], sd[p + "lin_query.bias"])
    v = F.linear(x, sd[p + "lin_value.weight"], sd[p + "lin_value.bias"])
    src, dst = edge_index[0], edge_index[1]
    msg = torch.sigmoid(k[dst] + q[src]) * v[src]
    out = torch.zeros_like(k).index_add_(0, dst, msg)
    if p + "lin_skip.weight" in sd:
        out = out + F.linear(x, sd[p + "lin_skip.weight"])
    if p + "bias" in sd:
        out = out + sd[p + "bias"]
    return out


def sage_conv(sd, p: str, x, edge_index, aggr: str = "mean"):
    """torch_geometric.nn.SAGEConv(in, out, aggr): lin_l(aggr_{j->i} x_j) + lin_r(x_i); GraphConv is the same form under the names lin_rel /
    lin_root with aggr = "add".  A node without incoming edges aggregates 0 (mean = sum / max(count, 1)).  UNPINNED restatement."""
    nbr, root = ("lin_l.", "lin_r.") if p + "lin_l.weight" in sd else ("lin_rel.", "lin_root.")
    src, dst = edge_index[0], edge_index[1]
    agg = torch.zeros_like(x).index_add_(0, dst, x[src])
    if aggr == "mean":
        cnt = torch.zeros(x.shape[0], dtype=x.dtype).index_add_(0, dst, torch.ones(dst.shape[0], dtype=x.dtype))
        agg = agg / cnt.clamp(min=1.0)[:, None]
    out = F.linear(agg, sd[p + nbr + "weight"], sd.get(p + nbr + "bias"))
    if p + root + "weight" in sd:
        out = out + F.linear(x, sd[p + root + "weight"])
    return out


def gcn_conv(sd, p: str, x, edge_index):
    """torch_geometric.nn.GCNConv(in, out): gcn_norm with add_self_loops (unit weights): deg = in-degree + 1, norm_ji = deg_j^-1/2 deg_i^-1/2;
    out_i = sum_{j->i, and i itself} norm_ji (W x_j) + bias.  UNPINNED restatement."""
    n = x.shape[0]
    keep = edge_index[0] != edge_index[1]                      # add_remaining_self_loops: a node keeps ONE loop of weight 1 whether it had one or not
    loops = torch.arange(n, dtype=edge_index.dtype)
    src, dst = torch.cat((edge_index[0][keep], loops)), torch.cat((edge_index[1][keep], loops))
    deg = torch.zeros(n, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.shape[0], dtype=x.dtype))
    dis = deg.pow(-0.5)
    norm = dis[src] * dis[dst]
    h = F.linear(x, sd[p + "lin.weight"])
    out = torch.zeros_like(h).index_add_(0, dst, norm[:, None] * h[src])
    return out + sd[p + "bias"] if p + "bias" in sd else out


def gat_conv(sd, p: str, x, edge_index, heads: int, negative_slope: float = 0.2):
    """torch_geometric.nn.GATConv(in, out / heads, heads, concat=True, add_self_loops=True) in eval mode: self loops of the input dropped, one per
    node appended; alpha = softmax over the incoming edges of leaky_relu(<x'_j, att_src> + <x'_i, att_dst>) per head
    (torch_geometric.utils.softmax: exp(e - max_i) / (sum_i + 1e-16)).  UNPINNED restatement."""
    n = x.shape[0]
    keep = edge_index[0] != edge_index[1]
    loops = torch.arange(n, dtype=edge_index.dtype)
    src, dst = torch.cat((edge_index[0][keep], loops)), torch.cat((edge_index[1][keep], loops))
    h = F.linear(x, sd[p + "lin.weight"]).view(n, heads, -1)
    a_src = (h * sd[p + "att_src"].view(1, heads, -1)).sum(-1)
    a_dst = (h * sd[p + "att_dst"].view(1, heads, -1)).sum(-1)
    e = F.leaky_relu(a_src[src] + a_dst[dst], negative_slope)                                  # [E, H]
    mx = torch.full((n, heads), -float("inf"), dtype=x.dtype).scatter_reduce(0, dst[:, None].expand(-1, heads), e, "amax", include_self=True)
    w = torch.exp(e - mx[dst])
    den = torch.zeros(n, heads, dtype=x.dtype).index_add_(0, dst, w)
    alpha = w / (den[dst] + 1e-16)
    out = torch.zeros_like(h).index_add_(0, dst, alpha[:, :, None] * h[src]).reshape(n, -1)
    return out + sd[p + "bias"] if p + "bias" in sd else out


def graph_conv(sd, p: str, x, edge_index, gnn_param: Optional[dict] = None):
    """The layer `gnn_conv` built (models/layer/gnn_conv.py:4-28), told apart by its parameter names; options a state dict does not carry
    (aggr, heads, negative_slope) come from `gnn_param`."""
    gp = gnn_param or {}
    if p + "lin_key.weight" in sd:
        return res_gated_graph_conv(sd, p, x, edge_index)
    if p + "lin_l.weight" in sd:
        return sage_conv(sd, p, x, edge_index, {"sum": "add"}.get(gp.get("aggr", "mean"), gp.get("aggr", "mean")))
    if p + "lin_rel.weight" in sd:
        return sage_conv(sd, p, x, edge_index, {"sum": "add"}.get(gp.get("aggr", "add"), gp.get("aggr", "add")))
    if p + "att_src" in sd:
        return gat_conv(sd, p, x, edge_index, int(gp["heads"]), float(gp.get("negative_slope", 0.2)))
    if p + "lin.weight" in sd:
        return gcn_conv(sd, p, x, edge_index)
    raise KeyError("no graph convolution under " + p)


def _tcn(sd, p: str, x):                                                             # TcnBlock, ugnet.py:55-76
    w = sd[p + "conv.weight"]
    pad = w.shape[-1] - 1
    out = F.conv2d(x, w, sd[p + "conv.bias"], padding=(1, pad))[:, :, :, :-pad]
    skip = F.conv2d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"]) if (p + "shortcut.weight") in sd else x
    return out + skip


def residual_block(sd, p: str, x, temb, edge_index, td_h: int, gnn_param: Optional[dict] = None):                      # ResidualBlock.forward, ugnet.py:115-135
    h = _tcn(sd, p + "tcn1.", x)
    h = h + F.conv2d(temb[:, :, None, None], sd[p + "t_conv.weight"], sd[p + "t_conv.bias"])
    h = _tcn(sd, p + "tcn2.", h)
    c = h.shape[1]
    h = F.layer_norm(h.transpose(1, 3), (1, c), sd[p + "norm.weight"], sd[p + "norm.bias"]).transpose(1, 3)
    h = F.conv2d(h, sd[p + "downsampling.weight"], sd[p + "downsampling.bias"], padding=(0, td_h // 2)).transpose(1, 3).squeeze(2)
    s = torch.relu(graph_conv(sd, p + "spatial.gnn.", h.reshape(h.shape[0], -1), edge_index, gnn_param))
    s = s.reshape(s.shape[0], td_h, -1).unsqueeze(2).transpose(1, 3)
    h = F.conv_transpose2d(s, sd[p + "upsampling.weight"], sd[p + "upsampling.bias"], padding=(0, td_h // 2))
    short = F.conv2d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"]) if (p + "shortcut.weight") in sd else x
    return h + short


def _steps(sd, prefix: str) -> List[Tuple[str, str]]:
    """Module sequence of `down` / `up` from the state-dict keys: ('res', prefix) for Down/UpBlock, ('conv', prefix) for
    Downsample / Upsample."""
    idx = sorted({int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)})
    out = []
    for i in idx:
        p = "%s%d." % (prefix, i)
        out.append(("res", p + "res.") if (p + "res.norm.weight") in sd else ("conv", p + "conv."))
    return out


def ugnet(sd, net_param: dict, x: torch.Tensor, t: torch.Tensor, x_masked: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """UGnet.forward (ugnet.py:252-295): x, x_masked [N, T, F], t [1] or [N] -> eps [N, T, F]."""
    td_h, d_h, gp = net_param["Td_h"], net_param["d_h"], net_param.get("gnn_param")
    h0 = torch.cat((x.unsqueeze(2).transpose(1, 3), x_masked.unsqueeze(2).transpose(1, 3)), dim=-1)
    xx = F.conv2d(h0, sd[NET + "x_proj.weight"], sd[NET + "x_proj.bias"])
    temb = time_embedding(t, d_h)
    hs = [xx]
    for kind, p in _steps(sd, NET + "down."):
        xx = residual_block(sd, p, xx, temb, edge_index, td_h, gp) if kind == "res" else \
            F.conv2d(xx, sd[p + "weight"], sd[p + "bias"], stride=(1, 2), padding=(0, 1))
        hs.append(xx)
    xx = residual_block(sd, NET + "middle.res1.", xx, temb, edge_index, td_h, gp)
    xx = residual_block(sd, NET + "middle.res2.", xx, temb, edge_index, td_h, gp)
    for kind, p in _steps(sd, NET + "up."):
        if kind == "conv":
            xx = F.conv_transpose2d(xx, sd[p + "weight"], sd[p + "bias"], stride=(1, 2), padding=(0, 1))
        else:
            xx = residual_block(sd, p, torch.cat((xx, hs.pop()), dim=1), temb, edge_index, td_h, gp)
    e = F.conv2d(xx, sd[NET + "out.0.weight"], sd[NET + "out.0.bias"])
    e = F.linear(e, sd[NET + "out.1.weight"], sd[NET + "out.1.bias"])
    return e.squeeze(2).transpose(1, 2)


# ---- schedule + posterior (diffusion_schedulers.py:39-125, graph_diffusion_model.py:46-73) -------------------------------
def gaussian_tables(T: int, schedule: str) -> Dict[str, np.ndarray]:
    if schedule == "linear":
        beta = np.linspace(1e-4, 2e-2, T)
    elif schedule == "quad":
        beta = np.linspace(1e-4 ** 0.5, 2e-2 ** 5, T) ** 2
    elif schedule == "cosine":
        cosn = lambda t: np.cos(math.pi * 0.5 * (t / T + 0.008) / (1 + 0.008)) ** 2
        ab = cosn(np.arange(0, T + 1, 1)) / cosn(0)
        beta = np.clip(1 - (ab[1:] / ab[:-1]), None, 0.999)
    else:
        raise AttributeError(schedule)
    alpha = np.concatenate((np.array([1.0]), 1 - beta))
    return {"beta": beta, "alpha": alpha, "alphabar": np.cumprod(alpha), "betabar": np.cumprod(beta)}


def inference_pair(kind: str, T: int, inference_T: int, i: int) -> Tuple[int, int]:
    if kind == "linear":
        t1 = T - int((float(i) / inference_T) * T)
        t2 = T - int((float(i + 1) / inference_T) * T)
    elif kind == "cosine":
        t1 = T - int(np.sin((float(i) / inference_T) * np.pi / 2) * T)
        t2 = T - int(np.sin((float(i + 1) / inference_T) * np.pi / 2) * T)
    else:
        raise ValueError(kind)
    return int(np.clip(t1, 1, T)), int(np.clip(t2, 0, T - 1))


def gaussian_posterior(tb, trick: str, target_t: int, t: int, pred, xt, noise: NoiseFn, record=None):
    atbar, atbar_target = tb["alphabar"][t], tb["alphabar"][target_t]
    if trick == "ddpm" or t <= 1:
        at = tb["alpha"][t]
        z = noise(tuple(xt.shape))
        if record is not None:
            record.append(z)
        atbar_prev = tb["alphabar"][t - 1]
        beta_tilde = tb["beta"][t - 1] * (1 - atbar_prev) / (1 - atbar)
        out = (1 / np.sqrt(at)).item() * (xt - ((1 - at) / np.sqrt(1 - atbar)).item() * pred)
        return out + np.sqrt(beta_tilde).item() * z
    if trick == "ddim":
        out = np.sqrt(atbar_target / atbar).item() * (xt - np.sqrt(1 - atbar).item() * pred)
        return out + np.sqrt(1 - atbar_target).item() * pred
    raise ValueError(trick)


def duplicate_edge_index(P: int, edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:      # graph_diffusion_model.py:77-84
    e = edge_index.reshape((2, 1, -1)) + torch.arange(0, P).view(1, -1, 1) * num_nodes
    return e.reshape((2, -1))


def evaluation_step(sd, net_param: dict, x: torch.Tensor, edge_index: torch.Tensor, noise: NoiseFn, record=None) -> torch.Tensor:
    """DiffSTG.evaluation_step (graph_diffusion_model.py:204-282): x [V, >= T_h, F] scaled -> [V, T, 1, P*S]."""
    T_h, T_p = net_param["T_h"], net_param["T_p"]
    P, S = net_param["parallel_sampling"], net_param["sequential_sampling"]
    trick = net_param.get("inference_trick") or "ddim"
    history = x[:, :T_h, :]
    V = x.shape[0]
    x_masked = torch.cat((history, torch.zeros(V, T_p, history.shape[2])), dim=1)
    if P > 1:
        edge_index = duplicate_edge_index(P, edge_index, V)
        x_masked = x_masked.repeat(P, 1, 1)
    tb = gaussian_tables(net_param["diffusion_steps"], net_param["diffusion_schedule"])
    steps = net_param["inference_diffusion_steps"]
    outs = []
    for _ in range(S):
        xt = noise(tuple(x_masked.shape))
        if record is not None:
            record.append(xt)
        for i in range(steps):
            t1, t2 = inference_pair(net_param["inference_schedule"], net_param["diffusion_steps"], steps, i)
            eps = ugnet(sd, net_param, xt.float(), torch.tensor([t1]).int().float(), x_masked, edge_index)
            xt = gaussian_posterior(tb, trick, t2, t1, eps, xt, noise, record)
        outs.append(xt.float())
    pred = torch.cat(outs, dim=0).reshape(P * S, -1, T_h + T_p, 1)
    return pred.permute(1, 2, 3, 0)
