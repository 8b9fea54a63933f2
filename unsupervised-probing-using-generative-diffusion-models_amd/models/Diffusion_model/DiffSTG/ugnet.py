"""Weight holders of DiffSTG's UGnet (mirrors reference DiffSTG/ugnet.py:55-250 in module / parameter NAMES and shapes, so a
reference checkpoint loads strictly; the graph convolution of a block is one of the holders of models/layer/gnn_conv.py, which carry
torch_geometric's parameter names).  Nothing here computes: the network runs in HIP kernels
(csrc/stg.hip) behind mpv_stg_*.  `layout()` describes the block sequence to the native loader."""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn


from ...layer.gnn_conv import ResGatedGraphConv, gnn_conv      # noqa: F401  (ResGatedGraphConv: the name older callers import from here)


class SpatialBlock(nn.Module):
    def __init__(self, c_in, c_out, gnn_name, gnn_param):
        super().__init__()
        self.gnn = gnn_conv(gnn_name, c_in, c_out, gnn_param)      # reference ugnet.py:37-39


class TcnBlock(nn.Module):
    def __init__(self, c_in, c_out, kernel_size):
        super().__init__()
        self.kernel_size = kernel_size
        self.padding = kernel_size - 1
        self.conv = nn.Conv2d(c_in, c_out, kernel_size=(3, kernel_size), padding=(1, self.padding), dilation=(1, 1))
        self.net = nn.Sequential(self.conv, nn.Identity(), nn.Identity())      # same conv registered twice, as in the reference
        self.shortcut = nn.Conv2d(c_in, c_out, kernel_size=(1, 1)) if c_in != c_out else None


class ResidualBlock(nn.Module):
    def __init__(self, c_in, c_out, T_in, net_param, kernel_size=3):
        super().__init__()
        self.c_in, self.c_out, self.T = c_in, c_out, T_in
        self.Td_h = net_param["Td_h"]
        self.tcn1 = TcnBlock(c_in, c_out, kernel_size)
        self.tcn2 = TcnBlock(c_out, c_out, kernel_size)
        self.shortcut = nn.Identity() if c_in == c_out else nn.Conv2d(c_in, c_out, (1, 1))
        self.t_conv = nn.Conv2d(net_param["d_h"], c_out, (1, 1))
        self.downsampling = nn.Conv2d(c_out, c_out, (1, T_in + 1), (1, 1), (0, self.Td_h // 2))
        self.upsampling = nn.ConvTranspose2d(c_out, c_out, (1, T_in + 1), (1, 1), (0, self.Td_h // 2))
        self.spatial = SpatialBlock(self.Td_h * c_out, self.Td_h * c_out, net_param["gnn_name"], net_param.get("gnn_param"))
        self.norm = nn.LayerNorm([1, c_out])


class DownBlock(nn.Module):
    def __init__(self, c_in, c_out, T_in, net_param):
        super().__init__()
        self.res = ResidualBlock(c_in, c_out, T_in, net_param)


class UpBlock(nn.Module):
    def __init__(self, c_in, c_out, T_in, net_param):
        super().__init__()
        self.res = ResidualBlock(c_in + c_out, c_out, T_in, net_param)


class Downsample(nn.Module):
    def __init__(self, c_in):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_in, kernel_size=(1, 3), stride=(1, 2), padding=(0, 1))


class Upsample(nn.Module):
    def __init__(self, c_in):
        super().__init__()
        self.conv = nn.ConvTranspose2d(c_in, c_in, (1, 4), (1, 2), (0, 1))


class MiddleBlock(nn.Module):
    def __init__(self, c_in, T_in, net_param):
        super().__init__()
        self.res1 = ResidualBlock(c_in, c_in, T_in, net_param)
        self.res2 = ResidualBlock(c_in, c_in, T_in, net_param)


class UGnet(nn.Module):
    def __init__(self, net_param) -> None:
        super().__init__()
        self.net_param = net_param
        self.d_h, self.T_p, self.T_h, self.F = net_param["d_h"], net_param["T_p"], net_param["T_h"], net_param["F"]
        self.T = self.T_p + self.T_h
        self.channel_multipliers = net_param["channel_multipliers"]
        self.n_blocks = net_param["n_blocks"]
        n_res = len(self.channel_multipliers)
        down, T_in = [], 2 * self.T
        out_c = in_c = self.d_h
        for i in range(n_res):
            out_c = in_c * self.channel_multipliers[i]
            for _ in range(self.n_blocks):
                down.append(DownBlock(in_c, out_c, T_in, net_param))
                in_c = out_c
            if i < n_res - 1:
                down.append(Downsample(in_c))
                T_in = math.floor((T_in - 1) / 2 + 1)
        self.down = nn.ModuleList(down)
        self.middle = MiddleBlock(out_c, T_in, net_param)
        up, in_c = [], out_c
        for i in reversed(range(n_res)):
            out_c = in_c
            for _ in range(self.n_blocks):
                up.append(UpBlock(in_c, out_c, T_in, net_param))
            out_c = in_c // self.channel_multipliers[i]
            up.append(UpBlock(in_c, out_c, T_in, net_param))
            in_c = out_c
            if i > 0:
                up.append(Upsample(in_c))
                T_in = T_in * 2
        self.up = nn.ModuleList(up)
        assert T_in == 2 * self.T, "T_in should be equal to 2*T"
        self.x_proj = nn.Conv2d(self.F, self.d_h, (1, 1))
        self.out = nn.Sequential(nn.Conv2d(self.d_h, self.F, (1, 1)), nn.Linear(2 * self.T, self.T))

    def layout(self) -> List[dict]:
        """Execution order of UGnet.forward (reference :252-295) as the native loader wants it: a list of steps
        {"kind": "res" | "down" | "up", "module": ..., "pop": bool (an up-path res block concatenates the popped skip)}."""
        steps = []
        for m in self.down:
            steps.append({"kind": "res", "module": m.res, "pop": False, "push": True} if isinstance(m, DownBlock)
                         else {"kind": "down", "module": m, "push": True})
        steps.append({"kind": "res", "module": self.middle.res1, "pop": False, "push": False})
        steps.append({"kind": "res", "module": self.middle.res2, "pop": False, "push": False})
        for m in self.up:
            steps.append({"kind": "up", "module": m} if isinstance(m, Upsample) else {"kind": "res", "module": m.res, "pop": True, "push": False})
        return steps
