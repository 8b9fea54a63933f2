#!/usr/bin/env python3
"""Time the TMDM sampler on a C2-shaped batch (W x 100 nodes x 100 positions x 32 samples, T = 50) in both GEMM modes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpv_amd import _native
from mpv_amd.models.Diffusion_model.TMDM.tmdm_diffusion_utils import tmdm_coef_rows, tmdm_tables

W = int(sys.argv[1]) if len(sys.argv) > 1 else 16
nf, T = int(os.environ.get("NF", 1)), 50
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
ctx = _native.Context(dev)
lin = lambda o, i: ((torch.rand(o, i, generator=g) * 2 - 1) / i ** 0.5, (torch.rand(o, generator=g) * 2 - 1) / i ** 0.5)
ws = [lin(128, 2 * nf), lin(128, 128), lin(128, 128)]
e4 = lin(nf, 128)
ctx.load_tmdm(T, nf, [w for w, _ in ws], [b for _, b in ws], [torch.rand(T + 1, 128, generator=g) for _ in range(3)], e4[0], e4[1],
              tmdm_coef_rows(tmdm_tables(T, "linear", 1e-4, 0.02)))
prior = (torch.randn(W, 100, 100, nf, generator=g) * 0.3).to(dev)
outs = {}
for gemm in ("f16x3", "f32"):
    ctx.sample_tmdm(prior, 32, 32, seed=1, gemm=gemm); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(3):
        outs[gemm] = ctx.sample_tmdm(prior, 32, 32, seed=5, gemm=gemm)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print("TMDM %-6s W=%d: %.2f ms/launch, %.3f ms/window = %.1f windows/s" % (gemm, W, ms, ms / W, W / ms * 1e3))
d = (outs["f16x3"].double() - outs["f32"].double()).abs()
print("f16x3 vs f32 on identical Philox noise: max %.3e, mean rel %.3e" % (float(d.max() / outs["f32"].abs().max()),
                                                                            float(d.mean() / outs["f32"].double().abs().mean())))
