"""CPU oracle for f(x), the non-stationary-transformer mean backbone, and for TMDM's
prior, which is the same network with a latent bottleneck between encoder and decoder.

TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() as the checker; the
product (package mpv_amd) never imports it.

PARITY UNPINNED.  The wiring is the reference's own file -- models/Diffusion_model/NsDiff/mu_backbone.py:12-183 (Projector,
Model.forward) and TMDM/tmdm_ns_transformer.py:8-172 -- but every layer it instantiates (`DataEmbedding`, `Encoder`,
`EncoderLayer`, `Decoder`, `DecoderLayer`, `AttentionLayer`, `DSAttention`) comes from torch-timeseries==0.1.10
(requirements.txt:10), which is absent from this image and cannot be fetched.  Those layers are restated here from the
published Non-stationary Transformer definition (Liu et al., NeurIPS 2022, section 3.2 and appendix):

  DataEmbedding      x -> Conv1d(c_in, d_model, k=3, circular padding, no bias)(x) + sinusoidal position table
                     (time-feature embedding off: the reference passes time_embed=False / x_mark=None)
  AttentionLayer     per-head projections Linear(d_model, d_model) for Q, K, V and an output Linear
  DSAttention        softmax(scale * (Q K^T * tau + delta)) V, scale = 1/sqrt(d_head), causal mask when mask_flag
  EncoderLayer       x = LN1(x + Attn(x));  x = LN2(x + W2 act(W1 x))          (W1, W2 are k=1 convolutions)
  DecoderLayer       x = LN1(x + SelfAttn(x, tau, delta=None));  x = LN2(x + CrossAttn(x, enc, tau, delta));
                     x = LN3(x + W2 act(W1 x))
  Encoder / Decoder  layers in order, a final LayerNorm, and for the decoder a Linear(d_model, c_out)

The state_dict names are the ones those classes give their parameters (value_embedding.tokenConv, position_embedding.pe,
attn_layers.N.attention.query_projection, layers.N.self_attention..., conv1 / conv2 / norm1..3, norm, projection), so a
`task_model: NsDiff` checkpoint's `cond_pred_model.*` entries address the same tensors.  No run of the real package is
available to confirm either the names or the arithmetic: every test built on this file says so.

Everything is functional: `forward(sd, cfg, x)` reads tensors out of a state_dict-like mapping.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

Tensors = Mapping[str, torch.Tensor]


def position_table(max_len: int, d_model: int) -> torch.Tensor:
    """Sinusoidal table [1, max_len, d_model]: even columns sin, odd columns cos of pos / 10000^(2i/d)."""
    pos = torch.arange(max_len, dtype=torch.float32).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float32) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def data_embedding(sd: Tensors, prefix: str, x: torch.Tensor) -> torch.Tensor:
    w = sd[prefix + "value_embedding.tokenConv.weight"]                      # [d_model, c_in, 3]
    xt = F.pad(x.transpose(1, 2), (1, 1), mode="circular")
    val = F.conv1d(xt, w).transpose(1, 2)
    return val + sd[prefix + "position_embedding.pe"][:, : x.shape[1]]


def ds_attention(q, k, v, tau, delta, causal: bool):
    """q [B, L, H, E], k / v [B, S, H, E]; tau [B, 1] | None; delta [B, S] | None."""
    B, L, H, E = q.shape
    S = k.shape[1]
    scores = torch.einsum("blhe,bshe->bhls", q, k)
    if tau is not None:
        scores = scores * tau.view(B, 1, 1, 1)
    if delta is not None:
        scores = scores + delta.view(B, 1, 1, S)
    if causal:
        mask = torch.triu(torch.ones(L, S, dtype=torch.bool), diagonal=1)
        scores = scores.masked_fill(mask, float("-inf"))
    a = torch.softmax(scores / math.sqrt(E), dim=-1)
    return torch.einsum("bhls,bshd->blhd", a, v)


def attention_layer(sd: Tensors, prefix: str, n_heads: int, xq, xkv, tau, delta, causal: bool):
    B, L, _ = xq.shape
    S = xkv.shape[1]
    q = F.linear(xq, sd[prefix + "query_projection.weight"], sd[prefix + "query_projection.bias"]).view(B, L, n_heads, -1)
    k = F.linear(xkv, sd[prefix + "key_projection.weight"], sd[prefix + "key_projection.bias"]).view(B, S, n_heads, -1)
    v = F.linear(xkv, sd[prefix + "value_projection.weight"], sd[prefix + "value_projection.bias"]).view(B, S, n_heads, -1)
    out = ds_attention(q, k, v, tau, delta, causal).reshape(B, L, -1)
    return F.linear(out, sd[prefix + "out_projection.weight"], sd[prefix + "out_projection.bias"])


def _act(name: str):
    return F.relu if name == "relu" else F.gelu


def _ffn(sd: Tensors, prefix: str, x, act):
    w1, b1 = sd[prefix + "conv1.weight"], sd[prefix + "conv1.bias"]           # [d_ff, d_model, 1]
    w2, b2 = sd[prefix + "conv2.weight"], sd[prefix + "conv2.bias"]
    return F.linear(act(F.linear(x, w1.squeeze(-1), b1)), w2.squeeze(-1), b2)


def _ln(sd: Tensors, prefix: str, x):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"])


def encoder(sd: Tensors, prefix: str, cfg, x, tau, delta):
    act = _act(cfg.activation)
    for i in range(cfg.e_layers):
        p = "{}attn_layers.{}.".format(prefix, i)
        x = _ln(sd, p + "norm1.", x + attention_layer(sd, p + "attention.", cfg.n_heads, x, x, tau, delta, False))
        x = _ln(sd, p + "norm2.", x + _ffn(sd, p, x, act))
    return _ln(sd, prefix + "norm.", x)


def decoder(sd: Tensors, prefix: str, cfg, x, cross, tau, delta):
    act = _act(cfg.activation)
    for i in range(cfg.d_layers):
        p = "{}layers.{}.".format(prefix, i)
        x = _ln(sd, p + "norm1.", x + attention_layer(sd, p + "self_attention.", cfg.n_heads, x, x, tau, None, True))
        x = _ln(sd, p + "norm2.", x + attention_layer(sd, p + "cross_attention.", cfg.n_heads, x, cross, tau, delta, False))
        x = _ln(sd, p + "norm3.", x + _ffn(sd, p, x, act))
    x = _ln(sd, prefix + "norm.", x)
    return F.linear(x, sd[prefix + "projection.weight"], sd[prefix + "projection.bias"])


def projector(sd: Tensors, prefix: str, n_layers: int, x_raw, stats):
    """mu_backbone.py:12-41: circular k=3 convolution ACROSS FEATURES with the time axis as channels, concatenated with the
    statistics row, then an MLP whose last Linear has no bias."""
    B = x_raw.shape[0]
    w = sd[prefix + "series_conv.weight"]                                     # [1, seq_len, 3]
    h = F.conv1d(F.pad(x_raw, (1, 1), mode="circular"), w)                    # [B, 1, E]
    h = torch.cat([h, stats], dim=1).reshape(B, -1)
    for i in range(n_layers):
        h = F.relu(F.linear(h, sd["{}backbone.{}.weight".format(prefix, 2 * i)], sd["{}backbone.{}.bias".format(prefix, 2 * i)]))
    return F.linear(h, sd["{}backbone.{}.weight".format(prefix, 2 * n_layers)])


def forward(sd: Tensors, cfg, x_enc: torch.Tensor, latent: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """mu_backbone.py:150-183 (latent=False) / tmdm_ns_transformer.py:136-172 in eval mode (latent=True: the encoder output
    passes through z_mean then z_out; z_logvar does not reach the prediction).  x_enc [B, seq_len, F], already scaled.
    Returns (dec_out[:, -pred_len:], dec_out) with dec_out [B, label_len + pred_len, c_out]."""
    x_raw = x_enc
    mean = x_enc.mean(1, keepdim=True)
    x = x_enc - mean
    std = torch.sqrt(torch.var(x, dim=1, keepdim=True, unbiased=False) + 1e-5)
    x = x / std
    x_dec = torch.cat([x[:, -cfg.label_len:, :], torch.zeros(x.shape[0], cfg.pred_len, x.shape[2], dtype=x.dtype)], dim=1)
    tau = projector(sd, "tau_learner.", cfg.p_hidden_layers, x_raw, std).exp()           # [B, 1]
    delta = projector(sd, "delta_learner.", cfg.p_hidden_layers, x_raw, mean)             # [B, seq_len]
    enc = encoder(sd, "encoder.", cfg, data_embedding(sd, "enc_embedding.", x), tau, delta)
    if latent:
        z = F.linear(F.relu(F.linear(enc, sd["z_mean.0.weight"], sd["z_mean.0.bias"])), sd["z_mean.2.weight"], sd["z_mean.2.bias"])
        enc = F.linear(F.relu(F.linear(z, sd["z_out.0.weight"], sd["z_out.0.bias"])), sd["z_out.2.weight"], sd["z_out.2.bias"])
    dec = decoder(sd, "decoder.", cfg, data_embedding(sd, "dec_embedding.", x_dec), enc, tau, delta)
    dec = dec * std + mean
    return dec[:, -cfg.pred_len:, :], dec


def spatial_stage(sd: Tensors, cfg, enc: torch.Tensor, edge_index: torch.Tensor, return_features: bool = False):
    """Model_spatial.forward between encoder and decoder (NsDiff/mu_backbone.py:312-325, the reference's own lines): Conv2d(D, D, (1, T + 1),
    padding (0, fT_h // 2)) along time, reshape to [rows, fT_h * D], `spatial_layers` x relu(gnn(x, edge_index)), ConvTranspose2d back.
    The graph convolution is torch_geometric's ResGatedGraphConv (models/layer/gnn_conv.py:18-19; package absent): the restatement of
    oracle/diffstg_oracle.py, unpinned.  enc [rows, S, D], rows = whole graphs (edge_index addresses rows)."""
    from oracle.diffstg_oracle import graph_conv          # the layer f_gnn_name selected, told apart by its parameter names (f_gnn_param: aggr, heads, ...)
    p = cfg.fT_h // 2
    h = enc.unsqueeze(2).transpose(1, 3)                                                       # [rows, D, 1, S]
    h = F.conv2d(h, sd["downsampling.weight"], sd["downsampling.bias"], padding=(0, p)).transpose(1, 3).squeeze(2)      # [rows, fT_h, D]
    s = h.reshape(h.shape[0], -1)
    for i in range(cfg.spatial_layers):
        s = torch.relu(graph_conv(sd, "spatial_encoder.{}.gnn.".format(i), s, edge_index, getattr(cfg, "f_gnn_param", None)))
    feat = s.reshape(h.shape[0], cfg.fT_h, -1)
    u = feat.unsqueeze(2).transpose(1, 3)                                                      # [rows, D, 1, fT_h]
    out = F.conv_transpose2d(u, sd["upsampling.weight"], sd["upsampling.bias"], padding=(0, p)).transpose(1, 3).squeeze(2)   # [rows, S, D]
    return (out, feat) if return_features else out


def forward_spatial(sd: Tensors, cfg, x_enc: torch.Tensor, edge_index: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Model_spatial.forward (NsDiff/mu_backbone.py:296-340): `forward` with the spatial stage between encoder and decoder.  x_enc [V, seq_len, F]
    = the nodes of ONE graph, edge_index [2, E] (row 0 source, row 1 target).  PARITY UNPINNED like `forward`, plus the graph convolution."""
    x_raw = x_enc
    mean = x_enc.mean(1, keepdim=True)
    x = x_enc - mean
    std = torch.sqrt(torch.var(x, dim=1, keepdim=True, unbiased=False) + 1e-5)
    x = x / std
    x_dec = torch.cat([x[:, -cfg.label_len:, :], torch.zeros(x.shape[0], cfg.pred_len, x.shape[2], dtype=x.dtype)], dim=1)
    tau = projector(sd, "tau_learner.", cfg.p_hidden_layers, x_raw, std).exp()
    delta = projector(sd, "delta_learner.", cfg.p_hidden_layers, x_raw, mean)
    enc = encoder(sd, "encoder.", cfg, data_embedding(sd, "enc_embedding.", x), tau, delta)
    enc = spatial_stage(sd, cfg, enc, edge_index)
    dec = decoder(sd, "decoder.", cfg, data_embedding(sd, "dec_embedding.", x_dec), enc, tau, delta)
    dec = dec * std + mean
    return dec[:, -cfg.pred_len:, :], dec


def random_spatial_state(cfg, seed: int, max_len: int = 5000) -> Dict[str, torch.Tensor]:
    """`random_state` + the spatial stage's tensors under the reference's names (downsampling / upsampling: mu_backbone.py:268-275;
    spatial_encoder.N.gnn.*: torch_geometric's ResGatedGraphConv parameter names)."""
    sd = random_state(cfg, seed, max_len=max_len)
    g = torch.Generator().manual_seed(seed + 7919)
    D, C, K = cfg.d_model, cfg.fT_h * cfg.d_model, cfg.seq_len + 1
    sd["downsampling.weight"] = torch.randn(D, D, 1, K, generator=g) / math.sqrt(D * K)
    sd["downsampling.bias"] = 0.05 * torch.randn(D, generator=g)
    for i in range(cfg.spatial_layers):
        p = "spatial_encoder.{}.gnn.".format(i)
        for nm in ("lin_key", "lin_query", "lin_value"):
            sd[p + nm + ".weight"] = torch.randn(C, C, generator=g) / math.sqrt(C)
            sd[p + nm + ".bias"] = 0.05 * torch.randn(C, generator=g)
        sd[p + "lin_skip.weight"] = torch.randn(C, C, generator=g) / math.sqrt(C)
        sd[p + "bias"] = 0.05 * torch.randn(C, generator=g)
    sd["upsampling.weight"] = torch.randn(D, D, 1, K, generator=g) / math.sqrt(D * cfg.fT_h)
    sd["upsampling.bias"] = 0.05 * torch.randn(D, generator=g)
    return sd


def intermediates(sd: Tensors, cfg, x_enc: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Stage outputs for kernel-by-kernel checks."""
    mean = x_enc.mean(1, keepdim=True)
    x = x_enc - mean
    std = torch.sqrt(torch.var(x, dim=1, keepdim=True, unbiased=False) + 1e-5)
    x = x / std
    tau = projector(sd, "tau_learner.", cfg.p_hidden_layers, x_enc, std).exp()
    delta = projector(sd, "delta_learner.", cfg.p_hidden_layers, x_enc, mean)
    emb = data_embedding(sd, "enc_embedding.", x)
    enc = encoder(sd, "encoder.", cfg, emb, tau, delta)
    return dict(mean=mean, std=std, x_norm=x, tau=tau, delta=delta, enc_embed=emb, enc_out=enc)


def random_state(cfg, seed: int, latent: bool = False, max_len: int = 5000) -> Dict[str, torch.Tensor]:
    """A seeded state_dict of the right shapes (normal(0, fan_in^-1/2) matrices, small biases, LayerNorm weights near 1) for
    tests that need weights but no checkpoint."""
    g = torch.Generator().manual_seed(seed)
    D, Fd, dff = cfg.d_model, cfg.dataset_nf, cfg.d_ff
    sd: Dict[str, torch.Tensor] = {}

    def mat(name, *shape, fan=None):
        fan = fan or shape[-1]
        sd[name] = torch.randn(*shape, generator=g) / math.sqrt(fan)

    def vec(name, n, base=0.0, spread=0.05):
        sd[name] = base + spread * torch.randn(n, generator=g)

    def lin(prefix, n_out, n_in, bias=True):
        mat(prefix + "weight", n_out, n_in)
        if bias:
            vec(prefix + "bias", n_out)

    def ln(prefix):
        vec(prefix + "weight", D, 1.0)
        vec(prefix + "bias", D)

    def attn(prefix):
        for nm in ("query", "key", "value", "out"):
            lin("{}{}_projection.".format(prefix, nm), D, D)

    def ffn(prefix):
        mat(prefix + "conv1.weight", dff, D, 1, fan=D)
        vec(prefix + "conv1.bias", dff)
        mat(prefix + "conv2.weight", D, dff, 1, fan=dff)
        vec(prefix + "conv2.bias", D)

    for emb in ("enc_embedding.", "dec_embedding."):
        mat(emb + "value_embedding.tokenConv.weight", D, Fd, 3, fan=3 * Fd)
        sd[emb + "position_embedding.pe"] = position_table(max_len, D)
    for i in range(cfg.e_layers):
        p = "encoder.attn_layers.{}.".format(i)
        attn(p + "attention.")
        ffn(p)
        ln(p + "norm1.")
        ln(p + "norm2.")
    ln("encoder.norm.")
    for i in range(cfg.d_layers):
        p = "decoder.layers.{}.".format(i)
        attn(p + "self_attention.")
        attn(p + "cross_attention.")
        ffn(p)
        for k in (1, 2, 3):
            ln("{}norm{}.".format(p, k))
    ln("decoder.norm.")
    lin("decoder.projection.", Fd, D)
    dims = list(cfg.p_hidden_dims)
    for name, n_out in (("tau_learner.", 1), ("delta_learner.", cfg.seq_len)):
        mat(name + "series_conv.weight", 1, cfg.seq_len, 3, fan=3 * cfg.seq_len)
        n_in = 2 * Fd
        for i in range(cfg.p_hidden_layers):
            lin("{}backbone.{}.".format(name, 2 * i), dims[i], n_in)
            n_in = dims[i]
        lin("{}backbone.{}.".format(name, 2 * cfg.p_hidden_layers), n_out, n_in, bias=False)
    if latent:
        for name in ("z_mean", "z_logvar", "z_out"):
            lin(name + ".0.", D, D)
            lin(name + ".2.", D, D)
    return sd


def config(dataset_nf=1, windows=100, pred_len=100, d_model=512, n_heads=8, d_ff=256, e_layers=4, d_layers=2,
           activation="gelu", p_hidden_dims=(64, 64), p_hidden_layers=2, fT_h=0, spatial_layers=0) -> SimpleNamespace:
    """The shipped `task_model: NsDiff` YAMLs' f(x) hyper-parameters (ews_results/NsDiff_dataset/*/model_trained.yaml); fT_h / spatial_layers:
    Model_spatial's (no shipped YAML names them)."""
    return SimpleNamespace(dataset_nf=dataset_nf, seq_len=windows, windows=windows, label_len=windows // 2, pred_len=pred_len, d_model=d_model,
                           n_heads=n_heads, d_ff=d_ff, e_layers=e_layers, d_layers=d_layers, activation=activation,
                           p_hidden_dims=list(p_hidden_dims), p_hidden_layers=p_hidden_layers, fT_h=fT_h, spatial_layers=spatial_layers)
