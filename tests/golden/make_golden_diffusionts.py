#!/usr/bin/env python3
"""Golden vectors for the DiffusionTS oracle, recorded from the reference's own package.

Build container only.  `models.Diffusion_model.DiffusionTS` imports without stand-ins.  Two seeded small models (one
with the 1-tap and one with the 5-tap circular `combine_s`), for each: the state dict, transformer outputs at several t,
one `model_predictions`, a complete `fast_sample_infill` and `evaluation_step` with every normal draw recorded
(torch.randn AND torch.randn_like are tapped: the sampler uses both).

    python tests/golden/make_golden_diffusionts.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as G                       # noqa: E402


class Tap:
    def __enter__(self):
        self.draws = []
        self._r, self._rl = torch.randn, torch.randn_like

        def randn(*a, **k):
            z = self._r(*a, **k)
            self.draws.append(z.detach().clone())
            return z

        def randn_like(x, *a, **k):
            z = self._rl(x, *a, **k)
            self.draws.append(z.detach().clone())
            return z

        torch.randn, torch.randn_like = randn, randn_like
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._r, self._rl

    def flat(self):
        return np.concatenate([d.reshape(-1).numpy() for d in self.draws])


class Replay:
    """torch.randn / randn_like return the recorded draws again (cast to the requested dtype): the float64 re-run of a case."""
    def __init__(self, draws):
        self.draws = list(draws)

    def __enter__(self):
        self._r, self._rl = torch.randn, torch.randn_like
        nxt = lambda: self.draws.pop(0)
        torch.randn = lambda *a, **k: nxt().to(k.get("dtype") or torch.get_default_dtype())
        torch.randn_like = lambda x, *a, **k: nxt().to(x.dtype)
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randn_like = self._r, self._rl


def record_float64(out, tag, m, x, t, batch, draws):
    """The same case evaluated by the reference's own modules in float64 (weights, inputs and the recorded normal draws cast up):
    the yardstick for what a float32 implementation can be asked to reproduce -- the reference's float32 run is itself only
    within |ref32 - ref64| of the exact result (2.8e-4 elementwise on the C3-dimension model)."""
    import copy
    m64 = copy.deepcopy(m).double()
    for mod in m64.modules():                      # TrendBlock keeps its polynomial basis as a plain attribute, not a buffer
        if hasattr(mod, "poly_space"):
            mod.poly_space = mod.poly_space.double()
    torch.set_default_dtype(torch.float64)
    try:
        xin = x.double().clone().requires_grad_(True)
        tt = torch.full((x.shape[0],), t, dtype=torch.long)
        with torch.enable_grad():
            y = m64.model.output(xin, tt)
            (gx,) = torch.autograd.grad((y * torch.from_numpy(out["vjp_%s_gout" % tag]).double()).sum(), xin)
        out["vjp_%s_y64" % tag], out["vjp_%s_gx64" % tag] = y.detach().numpy(), gx.detach().numpy()
        with Replay(draws):
            outs, _ = m64.evaluation_step(batch.double().clone())
        out["eval_%s_out64" % tag] = outs.numpy()
    finally:
        torch.set_default_dtype(torch.float32)


def record_blocks_and_grads(out, tag, m, x, t):
    """Per-block forward outputs of Transformer.forward (forward hooks on the reference's own blocks) and the input gradient of
    model_output for a seeded upstream gradient, with the gradients that reach the encoder output and the decoder input."""
    tf = m.model.model
    acts, hooks = {}, []
    for i, blk in enumerate(tf.encoder.blocks):
        hooks.append(blk.register_forward_hook(lambda mod, inp, o, i=i: acts.__setitem__("enc%d" % i, o[0])))
    for i, blk in enumerate(tf.decoder.blocks):
        hooks.append(blk.register_forward_hook(lambda mod, inp, o, i=i: acts.__setitem__("dec%d" % i, o)))
    xin = x.clone().requires_grad_(True)
    tt = torch.full((x.shape[0],), t, dtype=torch.long)
    with torch.enable_grad():
        y = m.model.output(xin, tt)
        gen = torch.Generator().manual_seed(41)
        g_out = torch.randn(y.shape, generator=gen)
        enc_last = acts["enc%d" % (len(tf.encoder.blocks) - 1)]
        enc_last.retain_grad()
        (gx,) = torch.autograd.grad((y * g_out).sum(), xin, retain_graph=True)
        (g_enc,) = torch.autograd.grad((y * g_out).sum(), enc_last)
    for h in hooks:
        h.remove()
    out["blk_%s_t" % tag] = np.array(t, dtype=np.int64)
    for i in range(len(tf.encoder.blocks)):
        out["blk_%s_enc%d" % (tag, i)] = G.npy(acts["enc%d" % i])
    for i in range(len(tf.decoder.blocks)):
        xo, mean, trend, season = acts["dec%d" % i]
        out["blk_%s_dec%d_x" % (tag, i)], out["blk_%s_dec%d_mean" % (tag, i)] = G.npy(xo), G.npy(mean)
        out["blk_%s_dec%d_trend" % (tag, i)], out["blk_%s_dec%d_season" % (tag, i)] = G.npy(trend), G.npy(season)
    out["vjp_%s_y" % tag], out["vjp_%s_gout" % tag], out["vjp_%s_gx" % tag], out["vjp_%s_genc" % tag] = G.npy(y), G.npy(g_out), G.npy(gx), G.npy(g_enc)


def main():
    sys.meta_path.append(G._AbsentFinder())
    sys.path.insert(0, str(G.REF))
    from models.Diffusion_model.DiffusionTS.DiffusionTS_model import DiffusionTS_model
    out = {}
    cases = {"small": dict(windows=12, pred_len=8, dataset_nf=2, d_model=16, n_layer_enc=1, n_layer_dec=2, n_heads=2, timesteps=20,
                           diffusion_steps=6, n_z_samples=4, parallel_sample=2),
             "wide": dict(windows=40, pred_len=30, dataset_nf=1, d_model=16, n_layer_enc=1, n_layer_dec=1, n_heads=4, timesteps=30,
                          diffusion_steps=30, n_z_samples=2, parallel_sample=2)}
    for tag, par in cases.items():
        torch.manual_seed({"small": 5, "wide": 6}[tag])
        m = DiffusionTS_model(dict(par, device="cpu", scaler_type="StandardScaler")).eval()
        for k, v in m.state_dict().items():
            out["sd_%s_%s" % (tag, k)] = G.npy(v)
        out["cfg_%s" % tag] = np.array([par[k] for k in ("windows", "pred_len", "dataset_nf", "d_model", "n_layer_enc", "n_layer_dec",
                                                            "n_heads", "timesteps", "diffusion_steps", "n_z_samples", "parallel_sample")], dtype=np.int64)
        g = torch.Generator().manual_seed(9)
        L, nf = par["windows"] + par["pred_len"], par["dataset_nf"]
        x = torch.randn(3, L, nf, generator=g) * 0.7
        out["x_%s" % tag] = G.npy(x)
        with torch.no_grad():
            for t in (0, par["timesteps"] // 2, par["timesteps"] - 1):
                tt = torch.full((3,), t, dtype=torch.long)
                trend, season = m.model.model(x, tt)
                out["tf_%s_t%d_trend" % (tag, t)] = G.npy(trend)
                out["tf_%s_t%d_season" % (tag, t)] = G.npy(season)
            tt = torch.full((3,), par["timesteps"] // 3, dtype=torch.long)
            pn, xs = m.model.model_predictions(x, tt, clip_x_start=True)
            out["mp_%s_noise" % tag], out["mp_%s_x0" % tag] = G.npy(pn), G.npy(xs)
        batch = torch.randn(2, par["windows"], nf, generator=g) * 0.5
        out["batch_%s" % tag] = G.npy(batch)
        torch.manual_seed(77)
        with Tap() as tap:
            outs, by = m.evaluation_step(batch.clone())
        assert by is None
        out["eval_%s_out" % tag], out["eval_%s_noise" % tag] = G.npy(outs), tap.flat()
        out["eval_%s_ndraws" % tag] = np.array([d.numel() for d in tap.draws], dtype=np.int64)
        record_blocks_and_grads(out, tag, m, x, par["timesteps"] // 2)
        record_float64(out, tag, m, x, par["timesteps"] // 2, batch, tap.draws)
    # BASELINE configs[2] dimensions (d_model 64, 4 heads x 16, L = 200, 3 + 6 blocks): weights = seeded default init of the
    # build's own parameter holders (same names and shapes; loaded strictly into the reference model), so no state dict is stored
    sys.path.insert(0, str(HERE.parents[1]))
    import mpv_amd  # noqa: F401
    from mpv_amd.models.Diffusion_model.DiffusionTS.DiffusionTS_model import DiffusionTS_model as Mirror
    par = dict(windows=100, pred_len=100, dataset_nf=1, d_model=64, n_layer_enc=3, n_layer_dec=6, n_heads=4, timesteps=100, diffusion_steps=5,
               n_z_samples=2, parallel_sample=2)
    torch.manual_seed(11)
    mirror = Mirror(dict(par, device="cpu", scaler_type="StandardScaler"))
    m = DiffusionTS_model(dict(par, device="cpu", scaler_type="StandardScaler")).eval()
    m.load_state_dict(mirror.state_dict(), strict=True)
    g = torch.Generator().manual_seed(19)
    x = torch.randn(2, 200, 1, generator=g) * 0.7
    out["x_c3"] = G.npy(x)
    with torch.no_grad():
        for t in (3, 50, 97):
            trend, season = m.model.model(x, torch.full((2,), t, dtype=torch.long))
            out["tf_c3_t%d_trend" % t], out["tf_c3_t%d_season" % t] = G.npy(trend), G.npy(season)
    record_blocks_and_grads(out, "c3", m, x, 50)
    batch = torch.randn(2, 100, 1, generator=g) * 0.5
    out["batch_c3"] = G.npy(batch)
    torch.manual_seed(78)
    with Tap() as tap:
        outs, _ = m.evaluation_step(batch.clone())
    out["eval_c3_out"], out["eval_c3_noise"] = G.npy(outs), tap.flat()
    record_float64(out, "c3", m, x, 50, batch, tap.draws)
    np.savez_compressed(HERE / "diffusionts_cases.npz", **out)
    f = HERE / "diffusionts_cases.npz"
    print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
