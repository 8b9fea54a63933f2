#!/usr/bin/env python3
"""Golden vectors for NsDiff_spatial (NsDiff_model_spatial / NsDiff_net_spatial / NsDiff's UGnet), recorded
from the reference's own modules.

Build container only.  `NsDiff/ugnet.py`, `NsDiff_net.py` and `NsDiff_model.py` import torch_geometric / torch_timeseries at module level
(absent here), which the inert placeholder finder of make_golden.py satisfies.  Two pieces of arithmetic live in those packages:
  * ResGatedGraphConv inside every ResidualBlock: replaced by the restated layer of make_golden_diffstg.py (`_RGGC`) -- that ONE layer is
    parity unpinned, as for DiffSTG;
  * the mean backbone f(x) = `Model_spatial` (torch_timeseries encoder / decoder classes): cannot run, so `cond_pred_model` is a stub that
    returns a SEEDED y_0_hat tensor, which is recorded.  evaluation_step treats f(x) as an input on this path.
Everything else that is recorded -- the schedule tables of NsDiff_net_spatial, UGnet.forward (x_proj over cat(y_t, y0hat, g), TCN blocks,
LayerNorm, k = T+1 down/up-sampling convolutions, U-Net wiring, out conv + Linear(T, T), the eps and softplus sigma heads), p_sample,
p_sample_t_1to0, the sigma head g(x), and NsDiff_model_spatial.evaluation_step with every normal draw (including its node-major row
tiling against the copy-major duplicated edge index) -- is the reference's code.

    python tests/golden/make_golden_nsdiff_spatial.py
"""
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden as G                                  # noqa: E402
from make_golden_diffstg import _RGGC, ring_graph        # noqa: E402
from make_golden_diffusionts import Tap                  # noqa: E402

CASES = {
    "tiny": dict(dataset_nf=1, windows=16, pred_len=8, rolling_length=8, Td_h=4, channel_multipliers=[2, 2], d_h=4, n_blocks=1, diffusion_steps=10,
                 n_z_samples=4, parallel_sample=2, V=6, extra=3, seed=3),
    "f2": dict(dataset_nf=2, windows=20, pred_len=12, rolling_length=6, Td_h=4, channel_multipliers=[2], d_h=4, n_blocks=1, diffusion_steps=6,
               n_z_samples=3, parallel_sample=3, V=5, extra=2, seed=5),
    "wide": dict(dataset_nf=1, windows=100, pred_len=100, rolling_length=50, Td_h=10, channel_multipliers=[2, 2], d_h=4, n_blocks=2, diffusion_steps=5,
                 n_z_samples=2, parallel_sample=2, V=8, extra=6, seed=4),
}


def net_param(par):
    keys = ("dataset_nf", "windows", "pred_len", "rolling_length", "Td_h", "channel_multipliers", "d_h", "n_blocks", "diffusion_steps",
            "n_z_samples", "parallel_sample")
    return dict({k: par[k] for k in keys}, device="cpu", scaler_type="StandardScaler", task_model="NsDiff_spatial", diffusion_schedule="linear",
                beta_start=1e-4, beta_end=0.02, load_pretrain=False, freeze_pretrain=False, gnn_name="ResGatedGraphConv",
                gnn_param={"bias": True, "root_weight": True})


class _SeededBackbone(torch.nn.Module):
    """Stands in for ns_Transformer.Model_spatial (absent third-party layers): returns the recorded y_0_hat."""
    def __init__(self, y0):
        super().__init__()
        self.y0 = y0

    def forward(self, batch_x, dec_inp, edge_index):
        assert batch_x.shape[0] == self.y0.shape[0]
        return self.y0, None


def build_reference(ref_model, ref_net, ref_g, ref_ug, net):
    """NsDiff_model_spatial with its own attributes, `model` (NsDiff_net_spatial) and sigma head; its constructor is bypassed only because it
    instantiates Model_spatial from placeholder classes.  evaluation_step and everything it calls run unmodified."""
    cls = ref_model.NsDiff_model_spatial
    m = cls.__new__(cls)
    torch.nn.Module.__init__(m)
    cfg = dict(net, seq_len=net["windows"], label_len=net["windows"] // 2)
    m.device, m.dataset_nf, m.windows, m.pred_len = "cpu", net["dataset_nf"], net["windows"], net["pred_len"]
    m.rolling_length, m.diffusion_steps, m.label_len, m.seq_len = net["rolling_length"], net["diffusion_steps"], cfg["label_len"], cfg["seq_len"]
    m.EPS, m.scaler, m.train_model_select = 10e-8, net["scaler_type"], "NsDiff_model"
    m.configs = SimpleNamespace(**cfg)
    m.register_buffer("scaler_mean", torch.zeros(m.dataset_nf))
    m.register_buffer("scaler_std", torch.ones(m.dataset_nf))
    m.model = ref_net.NsDiff_net_spatial(m.configs, "cpu")
    m.cond_pred_model_g = ref_g.SigmaEstimation(m.windows, m.pred_len, m.dataset_nf, 512, m.rolling_length).float()
    for mod in m.model.modules():
        if isinstance(mod, ref_ug.ResidualBlock):
            c = mod.Td_h * mod.c_out
            mod.spatial.gnn = _RGGC(c, c)
    return m.eval()


def main():
    sys.meta_path.append(G._AbsentFinder())
    sys.path.insert(0, str(G.REF))
    from models.Diffusion_model.NsDiff import NsDiff_model as RM, NsDiff_net as RN, g_backbone as RG, ugnet as RU, nsdiff_utils as U
    out = {}
    for tag, par in CASES.items():
        net = net_param(par)
        V, P, F, O, Tn = par["V"], par["parallel_sample"], par["dataset_nf"], par["pred_len"], par["diffusion_steps"]
        torch.manual_seed(par["seed"])
        m = build_reference(RM, RN, RG, RU, net)
        if tag == "wide":
            # C-sized dimensions: weights = seeded default init of the build's own parameter holders (same names and shapes, loaded
            # strictly into the reference model), so the state dict is not stored
            sys.path.insert(0, str(HERE.parents[1]))
            import mpv_amd  # noqa: F401
            from mpv_amd.models.Diffusion_model.NsDiff.NsDiff_model import NsDiff_model_spatial as Mirror
            torch.manual_seed(44)
            mirror = Mirror(dict(net), "NsDiff_model")
            with torch.no_grad():
                for prm in mirror.parameters():                     # default init leaves the k = T+1 convolutions tiny: spread them
                    if prm.dim() == 4 and prm.shape[-1] > 8:
                        prm.mul_(3.0)
            sd = {k: v for k, v in mirror.state_dict().items() if not k.startswith("cond_pred_model.")}
            m.load_state_dict(sd, strict=True)
        else:
            for k, v in m.state_dict().items():
                out["sd_%s_%s" % (tag, k)] = G.npy(v)
        edge = ring_graph(V, par["extra"], seed=V)
        out["edge_%s" % tag] = edge.numpy()
        out["cfg_%s" % tag] = np.array([V, par["windows"], O, par["rolling_length"], par["Td_h"], par["d_h"], par["n_blocks"], Tn, par["n_z_samples"], P, F,
                                         len(par["channel_multipliers"])], dtype=np.int64)
        net_obj = m.model
        for name in ("alphas", "one_minus_alphas_bar_sqrt", "alphas_cumprod", "alphas_cumprod_sum", "alphas_cumprod_prev", "alphas_cumprod_sum_prev",
                     "betas_tilde", "betas_bar", "betas_tilde_m_1", "betas_bar_m_1"):
            out["tab_%s_%s" % (tag, name)] = G.npy(getattr(net_obj, name))
        g = torch.Generator().manual_seed(17)
        N = V * P
        y = torch.randn(N, O, F, generator=g)
        y0 = torch.randn(N, O, F, generator=g) * 0.5
        gx = torch.rand(N, O, F, generator=g) * 0.5 + 0.05
        pedge = m.duplicate_edge_index(P, edge, V, "cpu") if P > 1 else edge
        net_obj.set_edge_index(pedge)
        out["y_%s" % tag], out["y0_%s" % tag], out["gx_%s" % tag] = G.npy(y), G.npy(y0), G.npy(gx)
        tabs = (net_obj.alphas, net_obj.one_minus_alphas_bar_sqrt, net_obj.alphas_cumprod, net_obj.alphas_cumprod_sum, net_obj.alphas_cumprod_prev,
                net_obj.alphas_cumprod_sum_prev, net_obj.betas_tilde, net_obj.betas_bar, net_obj.betas_tilde_m_1, net_obj.betas_bar_m_1)
        with torch.no_grad():
            for t in (0, Tn // 2, Tn - 1):
                eps, sig = net_obj(y, y0, gx, torch.tensor([t]))
                out["eps_%s_t%d" % (tag, t)], out["sig_%s_t%d" % (tag, t)] = G.npy(eps), G.npy(sig)
            t = Tn // 2
            torch.manual_seed(23)
            with Tap() as tap:
                y1 = U.p_sample(net_obj, y, y0, gx, y0, t, *tabs)
            out["step_%s_t" % tag], out["step_%s_out" % tag], out["step_%s_z" % tag] = np.int64(t), G.npy(y1), tap.flat()
            yl = U.p_sample_t_1to0(net_obj, y, y0, gx, y0, net_obj.one_minus_alphas_bar_sqrt, net_obj.alphas, net_obj.alphas_cumprod, net_obj.alphas_cumprod_sum,
                                   net_obj.alphas_cumprod_prev, net_obj.alphas_cumprod_sum_prev, net_obj.betas_tilde, net_obj.betas_bar,
                                   net_obj.betas_tilde_m_1, net_obj.betas_bar_m_1)
            out["last_%s_out" % tag] = G.npy(yl)
            # evaluation_step: f(x) stubbed with a seeded tensor, g(x) the reference's sigma head on the window
            batch = torch.cumsum(torch.randn(V, par["windows"], F, generator=g) * 0.3, dim=1)
            y0_eval = torch.randn(V, O, F, generator=g) * 0.4
            m.cond_pred_model = _SeededBackbone(y0_eval)
            out["batch_%s" % tag], out["y0eval_%s" % tag] = G.npy(batch), G.npy(y0_eval)
            out["gxeval_%s" % tag] = G.npy(m.cond_pred_model_g(batch))
            data = types.SimpleNamespace(x=batch.clone(), edge_index=edge.clone(), num_nodes=V)
            torch.manual_seed(31)
            with Tap() as tap:
                outs, truth = m.evaluation_step(data)
            assert truth is None and tuple(outs.shape) == (V, O, F, par["n_z_samples"])
            out["eval_%s_out" % tag], out["eval_%s_noise" % tag] = G.npy(outs), tap.flat()
            print(tag, "eval out", tuple(outs.shape), "finite", bool(torch.isfinite(outs).all()), "draws", len(tap.draws))
    f = HERE / "nsdiff_spatial_cases.npz"
    np.savez_compressed(f, **out)
    print("%-28s %8.1f KiB" % (f.name, f.stat().st_size / 1024))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
