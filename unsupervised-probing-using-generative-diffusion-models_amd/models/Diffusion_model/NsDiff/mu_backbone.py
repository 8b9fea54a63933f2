"""f(x), the non-stationary-transformer mean backbone: weight holders + native call.

Mirrors the reference's `Model` / `Projector` (models/Diffusion_model/NsDiff/mu_backbone.py:12-183) in constructor arguments
(a `configs` namespace with dataset_nf, seq_len, label_len, pred_len, d_model, n_heads, d_ff, e_layers, d_layers, activation,
p_hidden_dims, p_hidden_layers, ...) and in parameter names.  The encoder / decoder / attention / embedding containers stand in
for the torch-timeseries==0.1.10 classes the reference imports (`DataEmbedding`, `Encoder`, `EncoderLayer`, `Decoder`,
`DecoderLayer`, `AttentionLayer`): they carry the parameter names those classes are published with
(`value_embedding.tokenConv`, `position_embedding.pe`, `attn_layers.N.attention.query_projection`, `layers.N.self_attention`,
`conv1`, `norm1`, `projection`, ...), so a `task_model: NsDiff` checkpoint's `cond_pred_model.*` entries address the same
tensors.  That package is absent here, so neither the names nor the layer arithmetic can be confirmed against it: PARITY
UNPINNED (oracle/ns_transformer_oracle.py).

Nothing here computes the network.  `forward(batch_x, dec_inp)` runs csrc/nst.hip through mpv_nst_forward; the decoder input
the reference builds from `dec_inp` is only used for its SHAPE there (zeros_like, mu_backbone.py:158-159), so it is ignored.
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn


CONFIG_KEYS = ("d_model", "n_heads", "d_ff", "e_layers", "d_layers", "activation", "p_hidden_dims", "p_hidden_layers")


def _position_table(max_len: int, d_model: int) -> torch.Tensor:
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


class _Named(nn.Module):
    """A container whose only job is to give a child module its published attribute name."""

    def __init__(self, **children):
        super().__init__()
        for name, child in children.items():
            setattr(self, name, child)


class _PositionTable(nn.Module):
    def __init__(self, d_model, max_len=5000):
        super().__init__()
        self.register_buffer("pe", _position_table(max_len, d_model))


def _embedding(c_in, d_model):
    conv = nn.Conv1d(c_in, d_model, kernel_size=3, padding=1, padding_mode="circular", bias=False)
    nn.init.kaiming_normal_(conv.weight, mode="fan_in", nonlinearity="leaky_relu")
    return _Named(value_embedding=_Named(tokenConv=conv), position_embedding=_PositionTable(d_model))


def _attention(d_model):
    return _Named(query_projection=nn.Linear(d_model, d_model), key_projection=nn.Linear(d_model, d_model),
                  value_projection=nn.Linear(d_model, d_model), out_projection=nn.Linear(d_model, d_model))


def _encoder_layer(d_model, d_ff):
    return _Named(attention=_attention(d_model), conv1=nn.Conv1d(d_model, d_ff, 1), conv2=nn.Conv1d(d_ff, d_model, 1),
                  norm1=nn.LayerNorm(d_model), norm2=nn.LayerNorm(d_model))


def _decoder_layer(d_model, d_ff):
    return _Named(self_attention=_attention(d_model), cross_attention=_attention(d_model), conv1=nn.Conv1d(d_model, d_ff, 1),
                  conv2=nn.Conv1d(d_ff, d_model, 1), norm1=nn.LayerNorm(d_model), norm2=nn.LayerNorm(d_model), norm3=nn.LayerNorm(d_model))


class Projector(nn.Module):
    """MLP that learns the de-stationary factors (reference :12-41): same layer list, holder only."""

    def __init__(self, enc_in, seq_len, hidden_dims, hidden_layers, output_dim, kernel_size=3):
        super().__init__()
        self.series_conv = nn.Conv1d(seq_len, 1, kernel_size=kernel_size, padding=1, padding_mode="circular", bias=False)
        layers = [nn.Linear(2 * enc_in, hidden_dims[0]), nn.ReLU()]
        for i in range(hidden_layers - 1):
            layers += [nn.Linear(hidden_dims[i], hidden_dims[i + 1]), nn.ReLU()]
        layers += [nn.Linear(hidden_dims[-1], output_dim, bias=False)]
        self.backbone = nn.Sequential(*layers)
        self.hidden_layers = hidden_layers

    def tensors(self) -> List[torch.Tensor]:
        out = [self.series_conv.weight]
        for i in range(self.hidden_layers):
            out += [self.backbone[2 * i].weight, self.backbone[2 * i].bias]
        return out + [self.backbone[2 * self.hidden_layers].weight]


def _attention_tensors(a):
    out = []
    for p in (a.query_projection, a.key_projection, a.value_projection, a.out_projection):
        out += [p.weight, p.bias]
    return out


class Model(nn.Module):
    """Non-stationary Transformer f(x).  `latent=True` adds TMDM's z_mean / z_logvar / z_out heads (tmdm_ns_transformer.py:98-118)."""

    latent = False
    gemm_mode = "f16x3"      # Linear layers: "f16x3" (two-term f16 split, fp32-equivalent) | "f32" (exact fp32 matrix pipe)

    def __init__(self, configs, native=None, enc_in=None):
        super().__init__()
        if getattr(configs, "gemm_mode", "f16x3") == "f32":
            self.gemm_mode = "f32"
        self.pred_len, self.seq_len, self.label_len = configs.pred_len, configs.seq_len, configs.label_len
        self.output_attention = getattr(configs, "output_attention", False)
        if self.output_attention:
            raise NotImplementedError("output_attention=True returns the attention maps, which the native path does not materialise")
        nf = enc_in if enc_in is not None else configs.dataset_nf
        D, dff = configs.d_model, configs.d_ff
        self.enc_in, self.d_model, self.d_ff, self.n_heads = nf, D, dff, configs.n_heads
        self.activation = configs.activation
        if self.activation not in ("relu", "gelu"):
            raise ValueError("activation must be 'relu' or 'gelu'")
        self.enc_embedding = _embedding(nf, D)
        self.dec_embedding = _embedding(nf, D)
        self.encoder = _Named(attn_layers=nn.ModuleList([_encoder_layer(D, dff) for _ in range(configs.e_layers)]), norm=nn.LayerNorm(D))
        self.decoder = _Named(layers=nn.ModuleList([_decoder_layer(D, dff) for _ in range(configs.d_layers)]), norm=nn.LayerNorm(D),
                              projection=nn.Linear(D, nf, bias=True))
        self.p_hidden_dims = list(configs.p_hidden_dims)
        self.p_hidden_layers = int(configs.p_hidden_layers)
        self.tau_learner = Projector(nf, configs.seq_len, self.p_hidden_dims, self.p_hidden_layers, 1)
        self.delta_learner = Projector(nf, configs.seq_len, self.p_hidden_dims, self.p_hidden_layers, configs.seq_len)
        if self.latent:
            for name in ("z_mean", "z_logvar", "z_out"):
                setattr(self, name, nn.Sequential(nn.Linear(D, D), nn.ReLU(), nn.Linear(D, D)))
        object.__setattr__(self, "_native", native)

    # ---- what the library loads: (descriptor fields, tensors in include/mpv_hip.h's order) -----------------------
    def native_weight_list(self):
        rows = max(self.seq_len, self.label_len + self.pred_len)
        t = []
        for emb in (self.enc_embedding, self.dec_embedding):
            t += [emb.value_embedding.tokenConv.weight, emb.position_embedding.pe[0, :rows]]
        for l in self.encoder.attn_layers:
            t += _attention_tensors(l.attention)
            t += [l.conv1.weight.squeeze(-1), l.conv1.bias, l.conv2.weight.squeeze(-1), l.conv2.bias, l.norm1.weight, l.norm1.bias, l.norm2.weight, l.norm2.bias]
        t += [self.encoder.norm.weight, self.encoder.norm.bias]
        for l in self.decoder.layers:
            t += _attention_tensors(l.self_attention) + _attention_tensors(l.cross_attention)
            t += [l.conv1.weight.squeeze(-1), l.conv1.bias, l.conv2.weight.squeeze(-1), l.conv2.bias]
            for n in (l.norm1, l.norm2, l.norm3):
                t += [n.weight, n.bias]
        t += [self.decoder.norm.weight, self.decoder.norm.bias, self.decoder.projection.weight, self.decoder.projection.bias]
        t += self.tau_learner.tensors() + self.delta_learner.tensors()
        if self.latent:
            for seq in (self.z_mean, self.z_out):
                t += [seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias]
        fields = dict(enc_in=self.enc_in, c_out=self.enc_in, seq_len=self.seq_len, label_len=self.label_len, pred_len=self.pred_len,
                      d_model=self.d_model, n_heads=self.n_heads, d_ff=self.d_ff, e_layers=len(self.encoder.attn_layers),
                      d_layers=len(self.decoder.layers), activation=int(self.activation == "gelu"), p_layers=self.p_hidden_layers,
                      p_hidden=self.p_hidden_dims[: self.p_hidden_layers], latent=int(self.latent), pe_rows=rows)
        return self._extend_native(fields, t)

    def _extend_native(self, fields, tensors):
        return fields, tensors

    def _run(self, x_enc):
        if x_enc.dim() != 3 or x_enc.shape[1] != self.seq_len or x_enc.shape[2] != self.enc_in:
            raise ValueError("f(x) was built for windows [B, {}, {}], got {}".format(self.seq_len, self.enc_in, tuple(x_enc.shape)))
        if self._native is None:
            raise RuntimeError("the backbone is not attached to a native handle")
        ctx = self._native.context(self)
        series = x_enc.to(device=ctx.device, dtype=torch.float32).contiguous()
        return ctx.nst_forward(series, 1, 0, 1, full=True, gemm=self.gemm_mode)[0]          # [B, label_len + pred_len, F]

    def forward(self, x_enc, x_dec=None, enc_self_mask=None, dec_self_mask=None, dec_enc_mask=None):
        """(batch_x [B, seq_len, F], dec_inp) -> (dec_out[:, -pred_len:], dec_out)   (reference :150-183)."""
        if enc_self_mask is not None or dec_self_mask is not None or dec_enc_mask is not None:
            raise NotImplementedError("attention masks other than the decoder's causal mask are not part of this path")
        dec_out = self._run(x_enc)
        return dec_out[:, -self.pred_len:, :], dec_out


class SpatialBlock(nn.Module):
    """Holder of one graph convolution of Model_spatial (reference :43-51): `gnn` is the holder `gnn_conv` builds for f_gnn_name / f_gnn_param
    (models/layer/gnn_conv.py: torch_geometric's parameter names)."""

    def __init__(self, c_in, c_out, gnn_name="ResGatedGraphConv", gnn_param=None):
        super().__init__()
        from ...layer.gnn_conv import gnn_conv
        if c_in != c_out:
            raise ValueError("SpatialBlock keeps the feature width")
        self.gnn = gnn_conv(gnn_name, c_in, c_out, gnn_param)

    def tensors(self):
        """The layer's 8 slots of the C ABI (None where its kind has no tensor)."""
        return list(self.gnn.native_tensors())


class Model_spatial(Model):
    """Non-stationary Transformer with a graph stage between encoder and decoder: f(x) of NsDiff_spatial (reference :186-340).  Adds the
    reference's `downsampling` (Conv2d(D, D, (1, windows + 1), padding (0, fT_h // 2))), `spatial_encoder` (spatial_layers x SpatialBlock over
    fT_h * d_model features) and `upsampling` (ConvTranspose2d) under their names; configs additionally carries fT_h (even), spatial_layers,
    f_gnn_name, f_gnn_param.  `forward(batch_x, dec_inp, edge_index)` runs csrc/nst.hip's spatial mode: batch_x = the nodes of ONE graph."""

    def __init__(self, configs, native=None, enc_in=None):
        super().__init__(configs, native=native, enc_in=enc_in)
        self.T = getattr(configs, "windows", configs.seq_len)
        if self.T != self.seq_len:
            raise ValueError("Model_spatial: windows must equal seq_len (the k = windows + 1 convolution maps seq_len steps to fT_h)")
        self.fT_h, self.spatial_layers = int(configs.fT_h), int(configs.spatial_layers)
        if self.fT_h < 2 or self.fT_h % 2 or self.fT_h > self.seq_len:
            raise ValueError("fT_h must be even and in [2, seq_len]")
        D, C = self.d_model, self.fT_h * self.d_model
        gnn_name = getattr(configs, "f_gnn_name", "ResGatedGraphConv")
        self.spatial_encoder = nn.ModuleList([SpatialBlock(C, C, gnn_name, getattr(configs, "f_gnn_param", None)) for _ in range(self.spatial_layers)])
        self.downsampling = nn.Conv2d(D, D, (1, self.T + 1), (1, 1), (0, self.fT_h // 2))
        self.upsampling = nn.ConvTranspose2d(D, D, (1, self.T + 1), (1, 1), (0, self.fT_h // 2))

    def _extend_native(self, fields, tensors):
        t = list(tensors) + [self.downsampling.weight, self.downsampling.bias]
        for blk in self.spatial_encoder:
            t += blk.tensors()
        t += [self.upsampling.weight, self.upsampling.bias]
        g = self.spatial_encoder[0].gnn if self.spatial_layers else None
        return dict(fields, fT_h=self.fT_h, spatial_layers=self.spatial_layers, spatial_gnn=g.native_kind() if g is not None else 0,
                    spatial_heads=getattr(g, "heads", 1), spatial_slope=getattr(g, "negative_slope", 0.2)), t

    def _run_graph(self, x_enc, edge_index):
        if x_enc.dim() != 3 or x_enc.shape[1] != self.seq_len or x_enc.shape[2] != self.enc_in:
            raise ValueError("f(x) was built for windows [V, {}, {}], got {}".format(self.seq_len, self.enc_in, tuple(x_enc.shape)))
        if self._native is None:
            raise RuntimeError("the backbone is not attached to a native handle")
        ctx = self._native.context(self)
        ctx.nst_set_graph(edge_index, x_enc.shape[0])
        series = x_enc.to(device=ctx.device, dtype=torch.float32).contiguous()
        return ctx.nst_forward(series, 1, 0, 1, full=True, gemm=self.gemm_mode)[0]          # [V, label_len + pred_len, F]

    def forward(self, x_enc, x_dec=None, edge_index=None, enc_self_mask=None, dec_self_mask=None, dec_enc_mask=None):
        """(batch_x [V, seq_len, F], dec_inp, edge_index [2, E]) -> (dec_out[:, -pred_len:], dec_out)   (reference :296-340)."""
        if enc_self_mask is not None or dec_self_mask is not None or dec_enc_mask is not None:
            raise NotImplementedError("attention masks other than the decoder's causal mask are not part of this path")
        if edge_index is None:
            raise ValueError("Model_spatial.forward needs the graph's edge_index")
        dec_out = self._run_graph(x_enc, edge_index)
        return dec_out[:, -self.pred_len:, :], dec_out
