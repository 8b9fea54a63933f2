"""Weight holders of the DiffusionTS transformer (mirrors reference DiffusionTS/diffusionts_transformer.py:381-438 and
diffusionts_model_utils.py:66-202 in module / parameter NAMES only, so that reference checkpoints load strictly).

Nothing here computes the network: the forward and the input-gradient of the 9-block encoder/decoder run in HIP kernels
(csrc/dts_*.hip) behind mpv_dts_forward / mpv_dts_vjp / mpv_dts_sample.  What the host contributes are tables that depend
only on the weights and the timestep, evaluated with the reference's own fp32 torch ops so they are bit-identical:
  * the AdaLayerNorm modulation rows scale(t), shift(t) = Linear(SiLU(SinusoidalPosEmb(t)))   (model_utils.py:49-61, 187-202)
  * torch.fft.rfftfreq(L)                                                                      (transformer.py:69-73)
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn


class _ConvMLP(nn.Module):                       # Conv_MLP: Transpose, Conv1d(k=3, pad=1), Dropout -> "sequential.1.*"
    def __init__(self, in_dim, out_dim):
        super().__init__()
        self.sequential = nn.Sequential(nn.Identity(), nn.Conv1d(in_dim, out_dim, 3, stride=1, padding=1), nn.Identity())


class _PosEnc(nn.Module):
    def __init__(self, d_model, max_len):
        super().__init__()
        self.pe = nn.Parameter(torch.empty(1, max_len, d_model))
        nn.init.uniform_(self.pe, -0.02, 0.02)


class _AdaLN(nn.Module):
    def __init__(self, n_embd):
        super().__init__()
        self.linear = nn.Linear(n_embd, n_embd * 2)


class _Attention(nn.Module):
    def __init__(self, n_embd, cond=None):
        super().__init__()
        self.key = nn.Linear(cond or n_embd, n_embd)
        self.query = nn.Linear(n_embd, n_embd)
        self.value = nn.Linear(cond or n_embd, n_embd)
        self.proj = nn.Linear(n_embd, n_embd)


def _mlp(n_embd, times):
    return nn.Sequential(nn.Linear(n_embd, times * n_embd), nn.Identity(), nn.Linear(times * n_embd, n_embd), nn.Identity())


class _EncoderBlock(nn.Module):
    def __init__(self, n_embd, times):
        super().__init__()
        self.ln1 = _AdaLN(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn = _Attention(n_embd)
        self.mlp = _mlp(n_embd, times)


class _Trend(nn.Module):
    def __init__(self, n_channel, n_embd, n_feat):
        super().__init__()
        self.trend = nn.Sequential(nn.Conv1d(n_channel, 3, 3, padding=1), nn.Identity(), nn.Identity(),
                                   nn.Conv1d(n_embd, n_feat, 3, stride=1, padding=1))


class _DecoderBlock(nn.Module):
    def __init__(self, n_channel, n_feat, n_embd, times):
        super().__init__()
        self.ln1 = _AdaLN(n_embd)
        self.ln2 = nn.LayerNorm(n_embd)
        self.attn1 = _Attention(n_embd)
        self.attn2 = _Attention(n_embd, cond=n_embd)
        self.ln1_1 = _AdaLN(n_embd)
        self.trend = _Trend(n_channel, n_embd, n_feat)
        self.mlp = _mlp(n_embd, times)
        self.proj = nn.Conv1d(n_channel, n_channel * 2, 1)
        self.linear = nn.Linear(n_embd, n_feat)


class _Blocks(nn.Module):
    def __init__(self, blocks):
        super().__init__()
        self.blocks = nn.Sequential(*blocks)


class Transformer(nn.Module):
    def __init__(self, n_feat, n_channel, n_layer_enc, n_layer_dec, n_embd, n_heads, mlp_hidden_times=4, conv_params=None):
        super().__init__()
        if n_embd % n_heads != 0:
            raise AssertionError("n_embd must be divisible by n_heads")
        self.n_feat, self.n_channel, self.n_embd, self.n_heads = n_feat, n_channel, n_embd, n_heads
        self.n_layer_enc, self.n_layer_dec, self.mlp_hidden_times = n_layer_enc, n_layer_dec, mlp_hidden_times
        self.emb = _ConvMLP(n_feat, n_embd)
        self.inverse = _ConvMLP(n_embd, n_feat)
        if conv_params is None or conv_params[0] is None:            # reference transformer.py:401-407
            kernel_size, padding = (1, 0) if (n_feat < 32 and n_channel < 64) else (5, 2)
        else:
            kernel_size, padding = conv_params
        if kernel_size != 2 * padding + 1:
            raise ValueError("combine_s needs a 'same' circular convolution (kernel_size = 2*padding + 1)")
        self.combine_taps = kernel_size
        self.combine_s = nn.Conv1d(n_embd, n_feat, kernel_size=kernel_size, stride=1, padding=padding, padding_mode="circular", bias=False)
        self.combine_m = nn.Conv1d(n_layer_dec, 1, kernel_size=1, stride=1, padding=0, padding_mode="circular", bias=False)
        self.encoder = _Blocks([_EncoderBlock(n_embd, mlp_hidden_times) for _ in range(n_layer_enc)])
        self.pos_enc = _PosEnc(n_embd, n_channel)
        self.decoder = _Blocks([_DecoderBlock(n_channel, n_feat, n_embd, mlp_hidden_times) for _ in range(n_layer_dec)])
        self.pos_dec = _PosEnc(n_embd, n_channel)

    # ---- host tables ----------------------------------------------------------------------------------------------
    def ada_tables(self, timesteps: int) -> torch.Tensor:
        """[timesteps][n_enc + 2*n_dec][2*n_embd] fp32: row (t, a) = cat(scale, shift) of AdaLayerNorm a at timestep t,
        order enc i: ln1; dec i: ln1, ln1_1.  Same torch ops as the reference (SinusoidalPosEmb, SiLU, Linear)."""
        d = self.n_embd
        half = d // 2
        with torch.no_grad():
            t = torch.arange(timesteps, dtype=torch.long)
            freq = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
            e = t[:, None] * freq[None, :]
            emb = torch.nn.functional.silu(torch.cat((e.sin(), e.cos()), dim=-1))
            mods = [b.ln1 for b in self.encoder.blocks]
            for b in self.decoder.blocks:
                mods += [b.ln1, b.ln1_1]
            rows = [torch.nn.functional.linear(emb, m.linear.weight.detach().float().cpu(), m.linear.bias.detach().float().cpu())
                    for m in mods]
        return torch.stack(rows, dim=1).contiguous()

    def native_weight_list(self) -> List[torch.Tensor]:
        """Tensors in the order mpv_dts_load documents (include/mpv_hip.h)."""
        out = [self.emb.sequential[1].weight, self.emb.sequential[1].bias, self.inverse.sequential[1].weight,
               self.inverse.sequential[1].bias, self.combine_s.weight, self.combine_m.weight, self.pos_enc.pe, self.pos_dec.pe]
        for b in self.encoder.blocks:
            out += [b.ln2.weight, b.ln2.bias]
            for lin in (b.attn.key, b.attn.query, b.attn.value, b.attn.proj, b.mlp[0], b.mlp[2]):
                out += [lin.weight, lin.bias]
        for b in self.decoder.blocks:
            out += [b.ln2.weight, b.ln2.bias]
            for att in (b.attn1, b.attn2):
                for lin in (att.key, att.query, att.value, att.proj):
                    out += [lin.weight, lin.bias]
            out += [b.trend.trend[0].weight, b.trend.trend[0].bias, b.trend.trend[3].weight, b.trend.trend[3].bias]
            for lin in (b.mlp[0], b.mlp[2]):
                out += [lin.weight, lin.bias]
            out += [b.proj.weight, b.proj.bias, b.linear.weight, b.linear.bias]
        return [t.detach().float().cpu().contiguous() for t in out]
