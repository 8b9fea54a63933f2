#!/usr/bin/env python3
"""Generate tests/golden/tmdm_cases.npz by running the REFERENCE's TMDM sampler code.

Build container only (needs /root/reference).  Uses the same import arrangement as
``make_golden.py``: absent third-party packages resolve to inert placeholder modules.

What executes reference code end-to-end here: ``tmdm_model.ConditionalGuidedModel``,
``tmdm_diffusion_utils.{make_beta_schedule, p_sample, p_sample_t_1to0, p_sample_loop}``, the table
construction of ``TMDM.__init__`` and the tile / un-tile arithmetic of
``TMDM_model.evaluation_step`` (re-driven below with a seeded stand-in prior because the real prior,
ns-Transformer, is third-party arithmetic that is absent: SURVEY.md §8c).  ``TMDM.forward`` calls
``enc_embedding`` (absent ``DataEmbedding``) and then ignores its output (`cat_x & cat_y_pred`,
tmdm_model.py:39-41), so the driver below hands ``p_sample_loop`` an object whose call skips that
dead embedding and invokes the reference's ``diffussion_model`` directly.

    python tests/golden/make_golden_tmdm.py
"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as G                       # noqa: E402  (import machinery + NoiseTap)


class EpsOnly(torch.nn.Module):
    """Calls the reference's epsilon network the way TMDM.forward does (TMDM.py:94-98), minus the unused
    data embedding."""

    def __init__(self, tmdm):
        super().__init__()
        self.diffussion_model = tmdm.diffussion_model

    def forward(self, x, x_mark, y, y_t, y_0_hat, t):
        return self.diffussion_model(None, y_t, y_0_hat, t)


def main():
    sys.meta_path.append(G._AbsentFinder())
    sys.path.insert(0, str(G.REF))
    from models.Diffusion_model.TMDM import TMDM as tmdm_mod, tmdm_diffusion_utils as U
    from oracle import tmdm_oracle as O

    out = {}
    g = torch.Generator().manual_seed(17)
    yml = str(G.REF / "models/Diffusion_model/TMDM/tmdm.yml")

    def build(nf, steps, sd, schedule="linear"):
        cfg = types.SimpleNamespace(diffusion_config_dir=yml, timesteps=steps, beta_schedule=schedule,
                                    beta_start=1e-4, beta_end=0.02, enc_in=nf, CART_input_x_embed_dim=64,
                                    embed="fixed", freq="h", dropout=0.05)
        net = tmdm_mod.TMDM(cfg, "cpu")
        pre = O.EPS_NET[len("model."):]
        net.diffussion_model.load_state_dict({k[len(O.EPS_NET):]: v for k, v in sd.items()})
        return net.eval()

    # schedule tables
    for steps, kind in ((10, "linear"), (50, "linear"), (100, "linear"), (12, "cosine"), (8, "quad")):
        net = build(1, steps, O.random_state_dict(1, steps, seed=1), schedule=kind)
        out["sched_%s_%d_alphas" % (kind, steps)] = G.npy(net.alphas)
        out["sched_%s_%d_s" % (kind, steps)] = G.npy(net.one_minus_alphas_bar_sqrt)

    for tag, nf, steps, seed in (("f1t50", 1, 50, 31), ("f2t10", 2, 10, 32), ("f3t100", 3, 100, 33)):
        sd = O.random_state_dict(nf, steps, seed=seed)
        net = build(nf, steps, sd)
        model = EpsOnly(net)
        for k, v in sd.items():
            out["sd_%s_%s" % (tag, k)] = G.npy(v)
        y = torch.randn(6, 9, nf, generator=g)
        y0 = torch.randn(6, 9, nf, generator=g) * 0.4 + 0.1
        out.update({"in_%s_y" % tag: G.npy(y), "in_%s_y0" % tag: G.npy(y0)})
        with torch.no_grad():
            eps = [G.npy(net.diffussion_model(None, y, y0, torch.tensor([t]))) for t in range(steps)]
            out["eps_%s" % tag] = np.stack(eps)
            for t in (steps - 1, steps // 2, 1):
                torch.manual_seed(500 + t)
                with G.NoiseTap() as tap:
                    nxt = U.p_sample(model, None, None, y, y0, y0, t, net.alphas, net.one_minus_alphas_bar_sqrt)
                out["step_%s_t%d_z" % (tag, t)] = G.npy(tap.draws[0])
                out["step_%s_t%d_out" % (tag, t)] = G.npy(nxt)
            out["step_%s_final" % tag] = G.npy(U.p_sample_t_1to0(model, None, None, y, y0, y0, net.one_minus_alphas_bar_sqrt))
            torch.manual_seed(88)
            with G.NoiseTap() as tap:
                seq = U.p_sample_loop(model, None, None, y0, y0, steps, net.alphas, net.one_minus_alphas_bar_sqrt)
            out["loop_%s_noise" % tag] = np.stack([G.npy(d) for d in tap.draws])
            out["loop_%s_out" % tag] = G.npy(seq[steps])

            # evaluation_step's tile / loop / un-tile (tmdm_adapter.py:122-155) around a seeded prior
            B, label_len, pred_len, n_z, par = 3, 4, 8, 6, 3
            prior = torch.randn(B, label_len + pred_len, nf, generator=g) * 0.5
            torch.manual_seed(99)
            preds = []
            with G.NoiseTap() as tap:
                for _ in range(n_z // par):
                    tile = prior.repeat(par, 1, 1, 1).transpose(0, 1).flatten(0, 1)
                    seq = U.p_sample_loop(model, None, None, tile, tile, steps, net.alphas, net.one_minus_alphas_bar_sqrt)
                    gen = seq[steps].reshape(B, par, label_len + pred_len, nf)
                    preds.append(gen[:, :, -pred_len:, :])
            outs = torch.cat(preds, dim=1).permute(0, 2, 3, 1)
            out.update({"eval_%s_prior" % tag: G.npy(prior), "eval_%s_noise" % tag: tap.flat(),
                        "eval_%s_out" % tag: G.npy(outs),
                        "eval_%s_dims" % tag: np.array([B, label_len, pred_len, n_z, par], dtype=np.int64)})

    np.savez_compressed(HERE / "tmdm_cases.npz", **out)
    f = HERE / "tmdm_cases.npz"
    print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
