#!/usr/bin/env python3
"""Golden vectors for DiffusionTS with eta != 0 (the stochastic branch of fast_sample_infill and the sigma-weighted Langevin terms,
reference DiffusionTS.py:294-300, 392-395), recorded from the reference's own modules.  Build container only.

The model is the `small` case of make_golden_diffusionts.py (same seed, so the same weights: eta is not a parameter); only evaluation_step is
recorded, with every normal draw, for eta = 0.5 and eta = 1.

    python tests/golden/make_golden_diffusionts_eta.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as G                                  # noqa: E402
from make_golden_diffusionts import Tap                  # noqa: E402

PAR = dict(windows=12, pred_len=8, dataset_nf=2, d_model=16, n_layer_enc=1, n_layer_dec=2, n_heads=2, timesteps=20, diffusion_steps=6, n_z_samples=4,
           parallel_sample=2)


def main():
    sys.meta_path.append(G._AbsentFinder())
    sys.path.insert(0, str(G.REF))
    from models.Diffusion_model.DiffusionTS.DiffusionTS_model import DiffusionTS_model
    out = {}
    for tag, eta in (("eta05", 0.5), ("eta1", 1.0)):
        torch.manual_seed(5)
        m = DiffusionTS_model(dict(PAR, eta=eta, device="cpu", scaler_type="StandardScaler")).eval()
        # (the weights are those of `sd_small_*` in diffusionts_cases.npz: same seed, and eta is not a parameter)
        g = torch.Generator().manual_seed(9)
        L, nf = PAR["windows"] + PAR["pred_len"], PAR["dataset_nf"]
        torch.randn(3, L, nf, generator=g)
        batch = torch.randn(2, PAR["windows"], nf, generator=g) * 0.5
        out["batch"] = G.npy(batch)
        torch.manual_seed(77)
        with Tap() as tap:
            outs, by = m.evaluation_step(batch.clone())
        assert by is None and float(m.model.eta) == eta
        out["eval_%s_out" % tag], out["eval_%s_noise" % tag] = G.npy(outs), tap.flat()
        print(tag, "eta", eta, tuple(outs.shape), "draws", len(tap.draws), "finite", bool(torch.isfinite(outs).all()))
    f = HERE / "diffusionts_eta_cases.npz"
    np.savez_compressed(f, **out)
    print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
