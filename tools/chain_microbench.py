#!/usr/bin/env python3
"""Time the chain kernel alone (HIP events inside the library) on a fixed C2-shaped batch.
usage: MPV_LIB=build_variants/x.so python tools/chain_microbench.py [windows] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpv_amd import _native
from mpv_amd.models.Diffusion_model.NsDiff.NsDiff_net import nsdiff_tables, posterior_coef_rows

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nf = int(os.environ.get("NF", 1)); T = 50
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ctx = _native.Context(dev)
lin = lambda o, i: ((torch.rand(o, i, generator=g) * 2 - 1) / i ** 0.5, (torch.rand(o, generator=g) * 2 - 1) / i ** 0.5)
ws = [lin(128, 3 * nf), lin(128, 128), lin(128, 128)]
e4, es = lin(nf, 128), lin(nf, 128)
ctx.load_nsdiff(T, nf, [w for w, _ in ws], [b for _, b in ws], [torch.rand(T, 128, generator=g) for _ in range(3)],
                e4[0], e4[1], es[0], es[1], posterior_coef_rows(nsdiff_tables(T, "linear", 1e-4, 0.02)))
gx = (torch.rand(W, 100, 100, nf, generator=g) * 0.5 + 0.05).to(dev)
GEMM = os.environ.get("GEMM", "f16x3")
ctx.sample(gx, None, 32, 32, seed=1, gemm=GEMM); torch.cuda.synchronize()
ctx.timing_enable(True); ctx.timing()
for i in range(reps):
    out = ctx.sample(gx, None, 32, 32, seed=2 + i, gemm=GEMM)
ms, n = ctx.timing()
flop = W * 100 * 100 * 32 * T * 256 * (256 + 5 * nf)
print("%-28s %-6s W=%d  %.2f ms/launch  %.3f ms/window  %.1f TFLOP/s (%.1f%% of 157.3)  checksum %.6f" % (
    os.environ.get("MPV_LIB", "default"), GEMM, W, ms, ms / W, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, float(out.double().mean())))
if os.environ.get("ACC"):
    a = ctx.sample(gx[:2], None, 32, 32, seed=7, gemm="f16x3").double()
    b = ctx.sample(gx[:2], None, 32, 32, seed=7, gemm="f32").double()
    d = (a - b).abs()
    print("   f16x3 vs f32 (same Philox noise): max|d|/max|ref| %.3e   mean|d|/mean|ref| %.3e   nan %d/%d" % (
        float(d[~d.isnan()].max() / b[~b.isnan()].abs().max()), float(d[~d.isnan()].mean() / b[~b.isnan()].abs().mean()),
        int(a.isnan().sum()), int(b.isnan().sum())))
