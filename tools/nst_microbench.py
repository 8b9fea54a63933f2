#!/usr/bin/env python3
"""Time f(x) (csrc/nst.hip) on the BASELINE configs[1] shape: 181 windows x 100 nodes, windows = pred_len = 100, F = 1, the shipped
NsDiff YAMLs' backbone (d_model 512, 8 heads, d_ff 256, 4 encoder + 2 decoder layers).  Prints ms per call, windows/s and the
fp32 TFLOP/s of the algorithmic FLOPs (causal self-attention counted at half)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import types
import torch
from mpv_amd import _native
from mpv_amd.models.Diffusion_model.NsDiff import mu_backbone


def flops_per_row(S, Ld, D, dff, e_layers, d_layers):
    enc = 2 * S * D * 3 * D + 4 * S * S * D + 2 * S * D * D + 4 * S * D * dff
    dec = (2 * Ld * D * 3 * D + 2 * Ld * Ld * D + 2 * Ld * D * D) + (2 * Ld * D * D + 2 * S * D * 2 * D + 4 * Ld * S * D + 2 * Ld * D * D) + 4 * Ld * D * dff
    return e_layers * enc + d_layers * dec


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 181
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    gemm = sys.argv[3] if len(sys.argv) > 3 else "f16x3"
    dev = torch.device("cuda", 0)
    cfg = types.SimpleNamespace(dataset_nf=1, seq_len=100, label_len=50, pred_len=100, d_model=512, n_heads=8, d_ff=256, e_layers=4, d_layers=2,
                                activation="gelu", p_hidden_dims=[64, 64], p_hidden_layers=2)
    torch.manual_seed(0)
    f = mu_backbone.Model(cfg)
    ctx = _native.Context(dev)
    ctx.load_nst(f)
    V, step = 100, 5
    series = torch.randn(V, 100 + step * (W - 1), 1).cumsum(1).to(dev) * 0.1
    out = ctx.nst_forward(series, W, 0, step, gemm=gemm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = ctx.nst_forward(series, W, 0, step, gemm=gemm)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = flops_per_row(100, 150, 512, 256, 4, 2) * V * W
    print("f(x) C2 [%s]: %d windows x %d nodes in %.2f ms = %.1f windows/s, %.3f GFLOP/node, %.1f TFLOP/s fp32-equivalent (fp32 matrix peak 157.3), checksum %.6f" % (
        gemm, W, V, ms, W / ms * 1e3, fl / V / W / 1e9, fl / ms / 1e9, float(out.double().mean())))


if __name__ == "__main__":
    main()
