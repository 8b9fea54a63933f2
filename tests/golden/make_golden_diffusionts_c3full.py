#!/usr/bin/env python3
"""A BASELINE configs[2] window on the REAL sampling plan, run by the reference itself.

Build container only.  The reference's own `DiffusionTS_model.evaluation_step` (`DiffusionTS_model.py:72-109` ->
`DiffusionTS.py:277-310` fast_sample_infill -> `:359-407` langevin_fn with its K in {0, 1, 2, 3} thresholds) on the C3 model
(d_model 64, 4 heads, 3 + 6 blocks, 200 tokens, timesteps = 100) with `diffusion_steps = 100` -- 100 model_predictions + 128
Langevin forward / backward passes per chain -- under `torch.manual_seed`, at the largest batch this container's memory holds
(autograd keeps every block's activations AND attention matrices of the batch: about 20 MB per sequence):

  c3full   `--nodes` x N = 32 sequences in ONE chunk (default 25 x 32 = 800; the batch size enters the arithmetic through the
           Langevin loss normalisers, so it is part of the case).  Stored: the complete `outs` [nodes, pred_len, 1, 32], its MPV
           and pred_mean in float64, the history batch, the seed, a checksum of the state dict.  NOT stored: the 327 x 800 x 200
           draws -- the GPU test regenerates them with `noise_source = "torch_cpu"` under the same seed (209 MB).
  t100     a small model (d_model 16, 1 + 2 blocks, 20 tokens, F = 2) on the same 100 / 100 plan with every draw recorded.

    python tests/golden/make_golden_diffusionts_c3full.py [--nodes 25] [--threads 8] [--steps 100]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as G                       # noqa: E402
from make_golden_diffusionts import Tap       # noqa: E402

C3 = dict(windows=100, pred_len=100, dataset_nf=1, d_model=64, n_layer_enc=3, n_layer_dec=6, n_heads=4, timesteps=100)
SEED_WEIGHTS, SEED_HIST, SEED_RUN = 123, 77, 4321


def c3_history(nodes, windows):
    """The seeded history batch of tests/test_full_size_gpu.py::make_c3 (scaled range), first `nodes` rows of a 100-node draw."""
    g = torch.Generator().manual_seed(SEED_HIST)
    t = torch.linspace(0, 6.0, windows).view(1, -1, 1)
    phase = torch.rand(100, 1, 1, generator=g) * 6.28
    hist = (0.6 * torch.sin(t * (1.0 + 0.3 * torch.rand(100, 1, 1, generator=g)) + phase)
            + 0.05 * torch.randn(100, windows, 1, generator=g)).clamp(-1, 1)
    return hist[:nodes].contiguous()


def state_checksum(sd):
    return np.array([float(v.double().abs().sum()) for _, v in sorted(sd.items()) if v.dtype.is_floating_point], dtype=np.float64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=25)
    ap.add_argument("--nz", type=int, default=32)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--out", default=str(HERE / "diffusionts_c3full.npz"))
    ap.add_argument("--skip-small", action="store_true")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    sys.meta_path.append(G._AbsentFinder())
    sys.path.insert(0, str(G.REF))
    from models.Diffusion_model.DiffusionTS.DiffusionTS_model import DiffusionTS_model
    sys.path.insert(0, str(HERE.parents[1]))
    import mpv_amd  # noqa: F401
    from mpv_amd.models.models import diffusion_models
    out = {}

    # ---- t100: small model, the full 100 / 100 plan, all draws recorded ----------------------------------------------------
    if not args.skip_small:
        par = dict(windows=12, pred_len=8, dataset_nf=2, d_model=16, n_layer_enc=1, n_layer_dec=2, n_heads=2, timesteps=100,
                   diffusion_steps=100, n_z_samples=4, parallel_sample=2)
        torch.manual_seed(15)
        m = DiffusionTS_model(dict(par, device="cpu", scaler_type="StandardScaler")).eval()
        for k, v in m.state_dict().items():
            out["sd_t100_%s" % k] = G.npy(v)
        out["cfg_t100"] = np.array([par[k] for k in ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec", "n_heads",
                                                     "timesteps", "diffusion_steps", "n_z_samples", "parallel_sample")], dtype=np.int64)
        g = torch.Generator().manual_seed(29)
        batch = torch.randn(2, par["windows"], par["dataset_nf"], generator=g) * 0.5
        out["batch_t100"] = G.npy(batch)
        torch.manual_seed(79)
        with Tap() as tap:
            outs, _ = m.evaluation_step(batch.clone())
        out["eval_t100_out"], out["eval_t100_noise"] = G.npy(outs), tap.flat()
        print("t100: %d draws, out %s" % (len(tap.draws), tuple(outs.shape)), flush=True)

    # ---- c3full ----------------------------------------------------------------------------------------------------------
    net = dict(C3, device="cpu", scaler_type="StandardScaler", task_model="DiffusionTS", diffusion_steps=args.steps,
               n_z_samples=args.nz, parallel_sample=args.nz)
    torch.manual_seed(SEED_WEIGHTS)                     # the very construction of make_c3 / bench_dts: seeded default init of the mirror
    mirror = diffusion_models("DiffusionTS", dict(net))
    m = DiffusionTS_model(dict(net)).eval()
    m.load_state_dict(mirror.state_dict(), strict=True)
    hist = c3_history(args.nodes, C3["windows"])
    torch.manual_seed(SEED_RUN)
    t0 = time.perf_counter()
    outs, _ = m.evaluation_step(hist.clone())
    dt = time.perf_counter() - t0
    x = outs.double()
    mpv, pm = float(x.var(dim=-1, unbiased=False).mean()), float(x.mean())
    print("c3full: %d nodes x %d samples, %d steps: %.1f s on %d threads; MPV %.8g pred_mean %.8g" % (args.nodes, args.nz, args.steps, dt, args.threads, mpv, pm),
          flush=True)
    out["c3full_cfg"] = np.array([args.nodes, args.nz, args.steps, SEED_WEIGHTS, SEED_HIST, SEED_RUN], dtype=np.int64)
    out["c3full_hist"], out["c3full_out"] = G.npy(hist), G.npy(outs)
    out["c3full_mpv"], out["c3full_pred_mean"] = np.array(mpv), np.array(pm)
    out["c3full_state_checksum"] = state_checksum(m.state_dict())
    out["c3full_seconds_threads"] = np.array([dt, args.threads])
    np.savez_compressed(args.out, **out)
    f = Path(args.out)
    print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    main()
