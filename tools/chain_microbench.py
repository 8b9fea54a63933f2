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
    a = ctx.sample(gx[:2], None, 32, 32, seed=7, gemm=GEMM).double()
    b = ctx.sample(gx[:2], None, 32, 32, seed=7, gemm="f32").double()
    d = (a - b).abs()
    mpv_a, mpv_b = a.var(dim=-1, unbiased=False).mean(dim=(1, 2, 3)), b.var(dim=-1, unbiased=False).mean(dim=(1, 2, 3))
    print("   MPV rel err per window vs f32: %s" % ", ".join("%.2e" % float(v) for v in ((mpv_a - mpv_b).abs() / mpv_b)))
    print("   " + GEMM + " vs f32 (same Philox noise): max|d|/max|ref| %.3e   mean|d|/mean|ref| %.3e   nan %d/%d" % (
        float(d[~d.isnan()].max() / b[~b.isnan()].abs().max()), float(d[~d.isnan()].mean() / b[~b.isnan()].abs().mean()),
        int(a.isnan().sum()), int(b.isnan().sum())))
if os.environ.get("TRACE"):
    import ctypes as C
    lib = _native.load_library()
    buf = (C.c_uint64 * 16)()
    lib.mpv_debug_trace(buf, 1)
    ctx.sample(gx, None, 32, 32, seed=11, gemm="f16x3"); torch.cuda.synchronize()
    lib.mpv_debug_trace(buf, 0)
    steps = buf[14]
    names = ["act1 t0-2 (+L1 mfma)", "L2 t0 ks0-5 || act1 t3", "norm + L2 t0 ks6-7,b", "L2 t1-3 || act2 t0-2", "L3 t0 ks0-5 || act2 t3",
             "norm + L3 t0 ks6-7,b", "L3 t1-3 || act3 t0-2", "act3 t3 + norm", "heads + posterior", "loop head (noise, rows)"]
    tot = sum(buf[i] for i in range(10))
    print("   trace of block 0 wave 0: %d steps, %.0f memtime ticks/step; s_memtime runs at %.3f GHz (vs 100 MHz s_memrealtime)" % (
        steps, tot / steps, buf[12] / max(buf[13], 1) * 0.1))
    print("   wave lifetimes (100 MHz ticks): min %d max %d; block0/wave0 tiles total %d realtime ticks, %d shader ticks" % (buf[10], buf[15], buf[13], buf[12]))
    for i, nm in enumerate(names):
        print("     %-28s %8.1f ticks/step  %5.1f %%" % (nm, buf[i] / steps, 100.0 * buf[i] / tot))
