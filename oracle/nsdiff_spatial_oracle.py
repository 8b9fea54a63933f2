"""CPU oracle for NsDiff_spatial: NsDiff's graph U-Net epsilon / sigma network and
NsDiff_model_spatial.evaluation_step.

TEST INFRASTRUCTURE ONLY: imported by tests/ as the checker; the product (package mpv_amd) never imports it.

Parity status
  * PINNED against the reference's own code (tests/golden/make_golden_nsdiff_spatial.py): NsDiff_net_spatial's schedule tables, UGnet.forward
    of NsDiff/ugnet.py (x_proj over cat(y_t, y_0_hat, g), the block sequence, out conv + Linear(T, T), eps and softplus sigma heads),
    p_sample / p_sample_t_1to0 with that network, the sigma head g(x), and evaluation_step with every normal draw -- run with the
    reference's modules, their `spatial.gnn` replaced by the restated graph convolution and f(x) stubbed by a recorded tensor.
  * UNPINNED: `res_gated_graph_conv` (torch_geometric absent: oracle/diffstg_oracle.py) and f(x) = Model_spatial (an input here).

Paths are relative to models/Diffusion_model/NsDiff/ in the reference checkout.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import diffstg_oracle as STG
from . import nsdiff_oracle as NS

NET = "model.diffussion_model."          # (sic) NsDiff_net.py:255


def ugnet(sd: Dict[str, torch.Tensor], net_param: dict, y_t, y0hat, gx, t: torch.Tensor, edge_index):
    """UGnet.forward (ugnet.py:257-292): y_t, y0hat, gx [N, T, F], t [1] -> (eps, sigma) [N, T, F]."""
    td_h, d_h, gp = net_param["Td_h"], net_param["d_h"], net_param.get("gnn_param")
    x = torch.cat((y_t, y0hat, gx), dim=-1).unsqueeze(2).transpose(1, 3)             # (N, 3F, 1, T)   :262-263
    xx = F.conv2d(x, sd[NET + "x_proj.weight"], sd[NET + "x_proj.bias"])
    temb = STG.time_embedding(t, d_h)
    hs = [xx]
    for kind, p in STG._steps(sd, NET + "down."):
        xx = STG.residual_block(sd, p, xx, temb, edge_index, td_h, gp) if kind == "res" else \
            F.conv2d(xx, sd[p + "weight"], sd[p + "bias"], stride=(1, 2), padding=(0, 1))
        hs.append(xx)
    xx = STG.residual_block(sd, NET + "middle.res1.", xx, temb, edge_index, td_h, gp)
    xx = STG.residual_block(sd, NET + "middle.res2.", xx, temb, edge_index, td_h, gp)
    for kind, p in STG._steps(sd, NET + "up."):
        if kind == "conv":
            xx = F.conv_transpose2d(xx, sd[p + "weight"], sd[p + "bias"], stride=(1, 2), padding=(0, 1))
        else:
            xx = STG.residual_block(sd, p, torch.cat((xx, hs.pop()), dim=1), temb, edge_index, td_h, gp)
    e = F.conv2d(xx, sd[NET + "out.0.weight"], sd[NET + "out.0.bias"])
    e = F.linear(e, sd[NET + "out.1.weight"], sd[NET + "out.1.bias"])               # Linear(T, T) over the time axis
    h = e.squeeze(2).transpose(1, 2)                                                 # (N, T, d_h)    :289
    eps = F.linear(h, sd[NET + "lin4.weight"], sd[NET + "lin4.bias"])
    sig = F.softplus(F.linear(F.softplus(h), sd[NET + "sigma_lin.weight"], sd[NET + "sigma_lin.bias"]))     # :291 (both read the d_h-wide h)
    return eps, sig


def duplicate_edge_index(P: int, edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:      # NsDiff_model.py:783-790
    return (edge_index.reshape(2, 1, -1) + (torch.arange(P).view(1, -1, 1) * num_nodes)).reshape(2, -1)


def network(net_param: dict, edge_index: torch.Tensor):
    """The `net` callable oracle/nsdiff_oracle.py's step functions take."""
    return lambda sd, y, y0hat, gx, t: ugnet(sd, net_param, y, y0hat, gx, torch.tensor([t]), edge_index)


def evaluation_step(sd, net_param: dict, x: torch.Tensor, edge_index: torch.Tensor, y0hat: torch.Tensor, noise: NS.NoiseFn,
                    record: Optional[List[torch.Tensor]] = None) -> torch.Tensor:
    """NsDiff_model_spatial.evaluation_step (NsDiff_model.py:695-781): x [V, >= windows, F] scaled, y0hat [V, O, F] = f(x) -> [V, O, F, S].
    Rows are tiled node-major (row v*P + p, :756-763) while the duplicated edge index is copy-major (:751-753) -- as the reference has it;
    g(x) without EPS (:748)."""
    windows, O, nf = net_param["windows"], net_param["pred_len"], net_param["dataset_nf"]
    P, S, steps = int(net_param["parallel_sample"]), int(net_param["n_z_samples"]), int(net_param["diffusion_steps"])
    V = x.shape[0]
    tb = NS.schedule_tables(steps, net_param["diffusion_schedule"], net_param["beta_start"], net_param["beta_end"])
    gx = NS.sigma_head(sd, x[:, :windows, :], net_param["rolling_length"], O)
    net = network(net_param, duplicate_edge_index(P, edge_index, V) if P > 1 else edge_index)
    chunks = []
    for _ in range(S // P):
        y0_t, g_t = NS._tile_node_major(y0hat, P), NS._tile_node_major(gx, P)
        y = NS.sample_loop(sd, tb, y0_t, g_t, y0_t, steps, noise, record, net=net)
        chunks.append(y.reshape(V, P, O, nf))
    return torch.cat(chunks, dim=1).permute(0, 2, 3, 1)
