"""CPU tests: the DiffSTG oracle (oracle/diffstg_oracle.py) and the host-side schedule mirror against fixtures recorded from the
reference's own modules (tests/golden/make_golden_diffstg.py; the graph convolution inside them is the restated layer, see the
oracle's docstring: that one piece is parity-unpinned)."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from oracle import diffstg_oracle as O

T = torch.from_numpy


@pytest.fixture(scope="module")
def z():
    return np.load(GOLDEN / "diffstg_cases.npz")


def case(z, tag):
    V, T_h, T_p, Td_h, d_h, nb, Tn, it, P, S, cosine, ddpm, nres = [int(v) for v in z["cfg_%s" % tag]]
    net = dict(F=1, T_h=T_h, T_p=T_p, Td_h=Td_h, d_h=d_h, n_blocks=nb, channel_multipliers=[2] * nres, diffusion_steps=Tn,
               inference_diffusion_steps=it, parallel_sampling=P, sequential_sampling=S, inference_schedule="cosine" if cosine else "linear",
               inference_trick="ddpm" if ddpm else "ddim", diffusion_schedule="linear", loss_weight_schedule="constant", mask_ratio=0.0,
               scaler_type="StandardScaler", gnn_name="ResGatedGraphConv", gnn_param={"bias": True, "root_weight": True}, device="cpu",
               task_model="DiffSTG")
    return net, V


def state(z, tag, net):
    if tag == "c5":                        # seeded default init of the package's parameter holders (make_golden_diffstg.py)
        from mpv_amd.models.Diffusion_model.DiffSTG.graph_diffusion_model import DiffSTG
        torch.manual_seed(44)
        m = DiffSTG(dict(net))
        with torch.no_grad():
            for prm in m.parameters():
                if prm.dim() == 4 and prm.shape[-1] > 8:
                    prm.mul_(3.0)
            m.scaler_mean.fill_(0.2)
            m.scaler_std.fill_(1.3)
        return {k: v.detach().clone() for k, v in m.state_dict().items()}
    pre = "sd_%s_" % tag
    return {k[len(pre):]: T(z[k]) for k in z.files if k.startswith(pre)}


def test_tables_pairs_and_plan_match_reference(z):
    from mpv_amd.models.Diffusion_model.DiffSTG import diffusion_schedulers as DS
    for sched in ("linear", "cosine"):
        for Tn in (20, 200):
            tb, d = O.gaussian_tables(Tn, sched), DS.GaussianDiffusion(Tn, sched)
            for name in ("beta", "alpha", "alphabar"):
                assert np.array_equal(tb[name], z["tab_%s_%d_%s" % (sched, Tn, name)])
                assert np.array_equal(getattr(d, name), z["tab_%s_%d_%s" % (sched, Tn, name)])
    for kind in ("linear", "cosine"):
        for Tn, it in ((200, 20), (200, 50), (20, 7)):
            want = z["pairs_%s_%d_%d" % (kind, Tn, it)]
            assert np.array_equal(np.array([O.inference_pair(kind, Tn, it, i) for i in range(it)]), want)
            s = DS.InferenceSchedule(kind, Tn, it)
            assert np.array_equal(np.array([[int(v) for v in s(i)] for i in range(it)]), want)
            plan = DS.posterior_plan(DS.GaussianDiffusion(Tn, "linear"), kind, it, "ddim")
            assert np.array_equal(plan[:, 0].astype(int), want[:, 0]) and np.array_equal(plan[:, 5].astype(int), want[:, 1])
            assert bool((plan[:, 1] > 0).sum() == (want[:, 0] <= 1).sum())


@pytest.mark.parametrize("tag", ["tiny", "ddpm", "c5"])
def test_ugnet_posterior_and_evaluation_step_match_reference(z, tag):
    net, V = case(z, tag)
    sd = state(z, tag, net)
    x, xm, edge = T(z["x_%s" % tag]), T(z["xm_%s" % tag]), T(z["edge_%s" % tag])
    for t in (1, net["diffusion_steps"] // 2, net["diffusion_steps"]):
        eps = O.ugnet(sd, net, x, torch.tensor([t]).float(), xm, edge)
        assert torch.equal(eps, T(z["eps_%s_t%d" % (tag, t)]))
    tb = O.gaussian_tables(net["diffusion_steps"], "linear")
    pred = T(z["post_%s_pred" % tag])
    for t, tt in ((net["diffusion_steps"], net["diffusion_steps"] - 3), (1, 0)):
        zz = T(z["post_%s_t%d_z" % (tag, t)])
        post = O.gaussian_posterior(tb, net["inference_trick"], tt, t, pred, x, lambda shape: zz.reshape(shape))
        assert torch.equal(post, T(z["post_%s_t%d" % (tag, t)]))
    draws, pos = T(z["eval_%s_noise" % tag]), [0]

    def replay(shape):
        n = int(np.prod(shape))
        out = draws[pos[0]: pos[0] + n].reshape(shape)
        pos[0] += n
        return out

    outs = O.evaluation_step(sd, net, T(z["batch_%s" % tag]), edge, replay)
    assert pos[0] == draws.numel() and torch.equal(outs, T(z["eval_%s_out" % tag]))


def test_mirror_loads_reference_state_dict_strictly(z):
    from mpv_amd.models.models import diffusion_models
    net, V = case(z, "tiny")
    m = diffusion_models("DiffSTG", dict(net))
    m.load_state_dict(state(z, "tiny", net), strict=True)
    plan = m._plan()
    assert tuple(plan.shape) == (net["inference_diffusion_steps"], 6) and plan.dtype == torch.float64
    with pytest.raises(NotImplementedError):
        diffusion_models("DiffSTG", dict(net, gnn_name="PNAConv"))             # (no native kernel; SAGE / Graph / GCN / GAT have: test_gnn_variants.py)
