#!/usr/bin/env python3
"""Time the sigma-head kernel on the BASELINE configs[1] shape (181 windows x 100 nodes, windows=pred_len=100, rolling 50)
and on the SLBP shape (F=2, windows=pred_len=200, rolling 100, 2000 single-series windows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpv_amd import _native

dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
for (F, Tw, k, O, V, W, step, L) in ((1, 100, 50, 100, 100, 181, 5, 1000), (2, 200, 100, 200, 1, 2000, 10, 20200)):
    ctx = _native.Context(dev)
    H = 512
    lin = lambda o, i: ((torch.rand(o, i, generator=g) * 2 - 1) / i ** 0.5, (torch.rand(o, generator=g) * 2 - 1) / i ** 0.5)
    (w0, b0), (w1, b1), (w2, b2) = lin(H, Tw - k), lin(H, H), lin(O, H)
    ln = lambda: (torch.ones(F, H), torch.zeros(F, H))
    (l0w, l0b), (l1w, l1b) = ln(), ln()
    ctx.load_sigma(Tw, k, O, F, H, [w0, b0, l0w, l0b, w1, b1, l1w, l1b, w2, b2])
    series = torch.randn(V, L, F, generator=g).to(dev)
    ctx.sigma_head(series, W, 0, step); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        gx = ctx.sigma_head(series, W, 0, step)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    rows = W * V
    flop = rows * F * 2.0 * ((Tw - k) * H + H * H + H * O)
    print("sigma head F=%d windows=%d rolling=%d: %d rows in %.3f ms = %.2f M rows/s, %.2f TFLOP/s fp32, checksum %.6f" % (
        F, Tw, k, rows, ms, rows / ms / 1e3, flop / ms / 1e9, float(gx.double().mean())))
