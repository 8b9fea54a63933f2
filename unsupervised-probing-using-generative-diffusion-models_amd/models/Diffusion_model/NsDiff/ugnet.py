"""Weight holder of NsDiff_spatial's UGnet (mirrors reference NsDiff/ugnet.py:194-292 in module / parameter NAMES and shapes, so a
reference checkpoint loads strictly).  The blocks are the DiffSTG holders (the reference keeps two copies of the same block classes;
the gated graph convolution holds torch_geometric's ResGatedGraphConv parameters).  What differs from DiffSTG's UGnet:
  * the input is cat(y_t, y_0_hat, g) on the feature axis: x_proj = Conv2d(3F, d_h, 1x1) at time length T = pred_len (not 2T);
  * the resolutions halve with `T_in // 2` (:224);
  * the head: out = Conv2d(d_h, d_h, 1x1) + Linear(T, T), then lin4 (eps) and sigma_lin (softplus sigma) over the d_h channels.
Nothing here computes: the network runs in HIP kernels (csrc/stg.hip, engine mode 1) behind mpv_nss_*."""
from __future__ import annotations

import torch.nn as nn

from ..DiffSTG import ugnet as _blocks


class UGnet(nn.Module):
    def __init__(self, net_param) -> None:
        super().__init__()
        self.net_param = net_param                       # a SimpleNamespace, as in the reference
        cfg = dict(vars(net_param))
        self.d_h, self.T_p, self.T_h, self.F = net_param.d_h, net_param.pred_len, net_param.windows, net_param.dataset_nf
        self.T = self.T_p
        self.Td_h = net_param.Td_h
        self.channel_multipliers = net_param.channel_multipliers
        self.n_blocks = net_param.n_blocks
        n_res = len(self.channel_multipliers)
        down, T_in = [], self.T
        out_c = in_c = self.d_h
        for i in range(n_res):
            out_c = in_c * self.channel_multipliers[i]
            for _ in range(self.n_blocks):
                down.append(_blocks.DownBlock(in_c, out_c, T_in, cfg))
                in_c = out_c
            if i < n_res - 1:
                down.append(_blocks.Downsample(in_c))
                if T_in % 2:
                    raise ValueError("NsDiff_spatial UGnet: pred_len must stay even at every resolution (got {} at level {}): the reference's "
                                     "T_in // 2 (:224) and its stride-2 convolution disagree for odd lengths".format(T_in, i))
                T_in = T_in // 2
        self.down = nn.ModuleList(down)
        self.middle = _blocks.MiddleBlock(out_c, T_in, cfg)
        up, in_c = [], out_c
        for i in reversed(range(n_res)):
            out_c = in_c
            for _ in range(self.n_blocks):
                up.append(_blocks.UpBlock(in_c, out_c, T_in, cfg))
            out_c = in_c // self.channel_multipliers[i]
            up.append(_blocks.UpBlock(in_c, out_c, T_in, cfg))
            in_c = out_c
            if i > 0:
                up.append(_blocks.Upsample(in_c))
                T_in = T_in * 2
        self.up = nn.ModuleList(up)
        assert T_in == self.T, "T_in should be equal to T"
        self.x_proj = nn.Conv2d(3 * self.F, self.d_h, (1, 1))
        self.out = nn.Sequential(nn.Conv2d(self.d_h, self.d_h, (1, 1)), nn.Linear(self.T, self.T))
        self.lin4 = nn.Linear(self.d_h, self.F)
        self.sigma_lin = nn.Linear(self.d_h, self.F)

    layout = _blocks.UGnet.layout

    def forward(self, *args, **kwargs):
        raise RuntimeError("the epsilon / sigma network runs in the HIP library (mpv_nss_forward); call the owning model's evaluation_step")
